"""BASELINE config 5 (1024 pairs, T = 2000, band +-64, beam 5, thr 0.1): pairs/s of successive 1024-pair batches in
stream order and overlapping on the handle's internal streams (fcd_set_overlap) -- 1024 pairs put ONE wavefront on every
SIMD, so independent batches share the chip at little cost to each other.  python tools/duplex_overlap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from fast_ctc_decode_amd import _native as nat
import fast_ctc_decode_amd as fcd
from duplex_account import gen

T, w, B = 2000, 64, int(os.environ.get("PAIRS", "1024"))
REPS = int(os.environ.get("REPS", "8"))
i = np.arange(T)
env = np.stack([np.maximum(0, i - w), np.minimum(T, i + w)], 1).astype(np.uint64)
h = nat.default_handle()
x1, x2 = gen(B, T, 5, 4), gen(B, T, 5, 5)
envs = torch.from_numpy(np.broadcast_to(env, (B, T, 2)).copy().view(np.int64)).cuda()
for mode, name in ((0, "logsumexp"), (1, "max")):
    ref = None
    for ov in (0, 2, 3, 4):
        h.set_overlap(ov)
        keep = [fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, logadd_mode=mode) for _ in range(max(ov, 1) + 1)]
        h.overlap_join()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        keep = [fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, logadd_mode=mode) for _ in range(REPS)]
        h.overlap_join()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / REPS * 1e3
        got = keep[-1].cpu()
        if ref is None:
            ref = got
        same = bool((got.out_len == ref.out_len).all() and (got.labels == ref.labels).all())
        print("%s, %d pairs per call, overlap %d: %.2f ms per call = %.0f pairs/s%s" % (
            name, B, ov, ms, B / ms * 1e3, "" if same else "  RESULTS DIFFER"), flush=True)
        del keep
    h.set_overlap(0)
