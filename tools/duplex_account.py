"""Where does the duplex kernel's time go?  Cycle account of BASELINE config 5 (1024 pairs, T = 2000, band +-64)
per phase of the time step, both log-add modes (include/fcd.h: fcd_debug_set_duplex_profile).

    python tools/duplex_account.py [pairs] [T] [band]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import fast_ctc_decode_amd as fcd
from fast_ctc_decode_amd import _native as nat


def gen(B, T, N, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    x = torch.rand((B, T, N), generator=g, device="cuda", dtype=torch.float32)
    return x / torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    x1, x2 = gen(B, T, 5, 4), gen(B, T, 5, 5)
    i = np.arange(T)
    env = np.stack([np.maximum(0, i - w), np.minimum(T, i + w)], 1).astype(np.uint64)
    envs = torch.from_numpy(np.broadcast_to(env, (B, T, 2)).copy().view(np.int64)).cuda()
    h = nat.default_handle()
    names = ["envelope+extend", "lds_tiles", "expansion", "window_builds", "rank+next_beam"]
    # dependent latency of one LogSpace::add, both flavours (one wavefront, a chain of 4096 adds)
    lat = {}
    cyc = torch.zeros(64, dtype=torch.int64, device="cuda")
    sink = torch.zeros(64, dtype=torch.float32, device="cuda")
    for mode in (0, 1):
        for _ in range(2):
            h.check(h.lib.fcd_logadd_latency_probe_dev(h.ptr, 4096, mode, C.c_void_p(cyc.data_ptr()), C.c_void_p(sink.data_ptr())))
            torch.cuda.synchronize()
        lat[mode] = float(cyc.cpu().numpy().astype(np.float64).mean()) / 4096.0
    for mode, mname in ((0, "logsumexp"), (1, "max")):
        for _ in range(2):
            r = fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, logadd_mode=mode)
        torch.cuda.synchronize()
        plain_ms = h.last_kernel_ms()
        prof = torch.zeros((B, 16), dtype=torch.int32, device="cuda")
        h.check(h.lib.fcd_debug_set_duplex_profile(h.ptr, C.c_void_p(prof.data_ptr())))
        r2 = fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, logadd_mode=mode)
        torch.cuda.synchronize()
        stamped_ms = h.last_kernel_ms()
        h.check(h.lib.fcd_debug_set_duplex_profile(h.ptr, None))
        if not os.environ.get("FCD_ACCOUNT_NOCHECK"):  # (knock-out experiments produce wrong results on purpose)
            assert torch.equal(r.labels[:, :8], r2.labels[:, :8]) and torch.equal(r.out_len, r2.out_len)
        a = prof.cpu().numpy().astype(np.float64)
        cyc = a[:, :5] * 64.0
        steps = a[:, 7]
        per_step = (cyc / steps[:, None]).mean(0)
        rec = {"mode": mname, "pairs": B, "T": T, "band": w, "kernel_ms": plain_ms, "kernel_ms_with_stamps": stamped_ms,
               "cycles_per_step": {n: round(v, 1) for n, v in zip(names, per_step)},
               "cycles_per_step_total": round(float(per_step.sum()), 1),
               "new_nodes_per_step": round(float((a[:, 6] / steps).mean()), 2),
               "build_loop_iterations_per_step": round(float((a[:, 5] / steps).mean()), 1),
               "cycles_per_build_iteration": round(float((cyc[:, 3] / np.maximum(a[:, 5], 1)).mean()), 1),
               # The dependent-chain roofline: a new node's window is W + 1 sequential rows, each one LogSpace::add
               # behind the previous (the two chains of a row run on a pair of lanes); every step builds at least one
               # pass of new nodes.  bound = steps x rows x the add's dependent latency; frac = bound / kernel time.
               "chain_roofline": {
                   "bound": "dependent log-add chain", "cycles_per_dependent_logadd": round(lat[mode], 1),
                   "rows_per_step": w * 2 + 1, "steps": T,
                   "bound_cycles_per_step": round((w * 2 + 1) * lat[mode], 1),
                   "frac": round((w * 2 + 1) * lat[mode] / float(per_step.sum()), 4)},
               "sequential_extension_fraction": round(float((a[:, 8] / np.maximum(a[:, 10], 1)).mean()), 4),
               "entering_nodes_per_step": round(float((a[:, 9] / steps).mean()), 3),
               # finer stamps inside "envelope+extend" (the read-2 tile) and "rank+next_beam" (cycles per step)
               "inside": {n: round(float((a[:, 11 + k] * 64.0 / steps).mean()), 1) for k, n in enumerate(
                   ["read-2 tile load", "buffer hand-over + entering nodes' bounds and rings", "update_max rescans",
                    "child rows of the next beam", "beam sort + parents' bounds"])}}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
