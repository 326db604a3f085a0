"""The viterbi kernel's duration measured two ways IN ONE PROCESS: HIP events around every launch
(what bench.py's viterbi_roofline quotes) and -- when this script runs under `rocprofv3 --kernel-trace --stats` --
the profiler's own kernel durations of the very same launches (tools/viterbi_clock.sh puts the two side by side).

    python tools/viterbi_clock.py [launches] [f32|f16]

Prints one JSON line: per-launch event times (mean / min / max / median) and the GPU clocks before and after."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import fast_ctc_decode_amd as fcd


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=30).stdout
        return [l.strip() for l in out.splitlines() if "sclk" in l or "mclk" in l or "fclk" in l][:6]
    except Exception as e:  # (no rocm-smi: say so)
        return ["rocm-smi unavailable: %s" % e]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    half = len(sys.argv) > 2 and sys.argv[2] == "f16"
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    x = torch.rand((16384, bench.T, bench.N), generator=g, device=dev, dtype=torch.float32)
    x /= torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)
    if half:
        x = x.to(torch.float16)
    before = clocks()
    for _ in range(5):
        r = fcd.viterbi_search_batch_raw(x)
    torch.cuda.synchronize()
    h = r._handle
    per = []
    h.timing_reset()
    for _ in range(n):
        r = fcd.viterbi_search_batch_raw(x)
        torch.cuda.synchronize()
        per.append(h.last_kernel_ms())  # the C ABI's own event pair around this launch
    after = clocks()
    per = np.array(per)
    mean_L = float(r.out_len.float().mean())
    bytes_per_read = bench.T * bench.N * (2 if half else 4) + 5.0 * mean_L
    alg = 16384 * bytes_per_read
    print(json.dumps({
        "kernel": "viterbi_stream_kernel<5, %d>" % (1 if half else 0), "launches": n,
        "kernel_ms_events": {"mean": float(per.mean()), "min": float(per.min()), "max": float(per.max()),
                             "median": float(np.median(per))},
        "algorithmic_bytes_per_launch": alg,
        "frac_of_8TBs": {"from_mean": alg / (per.mean() * 1e-3) / 8e12, "from_min": alg / (per.min() * 1e-3) / 8e12},
        "clocks_before": before, "clocks_after": after}), flush=True)


if __name__ == "__main__":
    main()
