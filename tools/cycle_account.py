"""Per-block cycle account of the headline kernel (beam 5, N = 5, two reads per wavefront): runs
fcd_beam_search_profile_dev -- the same search with a shader-clock stamp after each block of the time step --
at several batch sizes and prints one JSON line per size.  The stamps wait for each block's results, so the
numbers are the blocks' DEPENDENT latencies as one wavefront sees them (plus what co-resident wavefronts
cost it), not overlapped time.

    python tools/cycle_account.py [B ...]          (default: 512 2048 4096 16384)
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import fast_ctc_decode_amd as fcd
from fast_ctc_decode_amd import _native as nat

BLOCKS = ["row fetch (+ loop)", "extensions + push", "numbering + record stores", "end tests + key + exact rank",
          "child-entry upkeep + eviction", "gather survivors", "reload + top + 2 divisions + state"]


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [512, 2048, 4096, 16384]
    T = 4000
    h = nat.default_handle(0)
    for B in sizes:
        g = torch.Generator(device="cuda")
        g.manual_seed(1)
        x = torch.rand((B, T, 5), generator=g, device="cuda")
        x = x / torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)
        ref = fcd.beam_search_batch_raw(x, 5, 0.1, True)
        torch.cuda.synchronize()
        plain_ms = ref._handle.last_kernel_ms()
        labels = torch.empty((B, T), dtype=torch.uint8, device="cuda")
        path = torch.empty((B, T), dtype=torch.int32, device="cuda")
        out_len = torch.zeros(B, dtype=torch.int32, device="cuda")
        status = torch.zeros(B, dtype=torch.int32, device="cuda")
        n_waves = (B + 1) // 2
        cyc = torch.zeros((n_waves, 8), dtype=torch.int32, device="cuda")
        st = x.stride()
        b = nat.Batch(x.data_ptr(), B, T, 1, 5, st[0], st[1], 0, st[2], None)
        res = nat.Result(labels.data_ptr(), path.data_ptr(), None, out_len.data_ptr(), status.data_ptr(), T)
        h.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            h.check(h.lib.fcd_beam_search_profile_dev(h.ptr, C.byref(b), 5, 0.1, 1, C.byref(res), cyc.data_ptr()))
        torch.cuda.synchronize()
        prof_ms = h.last_kernel_ms()
        mask = torch.arange(T, device="cuda")[None, :] < out_len[:, None]
        assert torch.equal(out_len, ref.out_len) and torch.equal(labels[mask], ref.labels[mask]) \
            and torch.equal(path[mask], ref.path[mask]), "instrumented != plain"
        c = cyc.cpu().numpy().astype(np.int64)
        steps = c[:, 7].astype(np.float64)
        per = c[:, :7] / steps[:, None]
        mean = per.mean(0)
        out = {"reads": B, "wavefronts_per_simd": n_waves / 1024.0, "plain_kernel_ms": plain_ms,
               "instrumented_kernel_ms": prof_ms, "cycles_per_step_total": float(mean.sum()),
               "cycles_per_step": {BLOCKS[j]: round(float(mean[j]), 1) for j in range(7)},
               "implied_clock_GHz": float(mean.sum() * T / (prof_ms * 1e-3) / 1e9)}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
