"""Timing of the alignment-band estimator at BASELINE config-5 scale: python tools/probe_envelope.py [pairs] [labels]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import fast_ctc_decode_amd as fcd
from test_gpu_envelope import warped_pair


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n_labels = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    rng = np.random.default_rng(1)
    base = [warped_pair(rng, n_labels, mutate=0.05)[:2] for _ in range(16)]
    T1 = max(p[0].shape[0] for p in base)
    T2 = max(p[1].shape[0] for p in base)
    X1 = np.zeros((B, T1, 5), np.float32)
    X2 = np.zeros((B, T2, 5), np.float32)
    l1 = np.zeros(B, np.int64)
    l2 = np.zeros(B, np.int64)
    for i in range(B):
        a, b = base[i % 16]
        X1[i, :a.shape[0]], X2[i, :b.shape[0]] = a, b
        l1[i], l2[i] = a.shape[0], b.shape[0]
    x1, x2 = torch.from_numpy(X1).cuda(), torch.from_numpy(X2).cuda()
    torch.cuda.synchronize()
    for _ in range(2):
        t0 = time.perf_counter()
        env = fcd.estimate_envelope_batch(x1, x2, 32, l1, l2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    h = fcd.api.nat.default_handle()
    print("estimate_envelope_batch: %d pairs T1=%d T2=%d: wall %.1f ms (align+build kernel %.1f ms)"
          % (B, T1, T2, dt * 1e3, h.last_kernel_ms()), flush=True)
    e = env.cpu().numpy().view(np.uint64)
    w = (e[0, :l1[0], 1] - e[0, :l1[0], 0]).astype(np.int64)
    print("window width of pair 0: mean %.1f max %d" % (w.mean(), w.max()))
    for name, envs in (("estimated band", env), ("full envelope", None)):
        t0 = time.perf_counter()
        r = fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, lengths_1=l1, lengths_2=l2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("beam_search_duplex, %s: %.1f ms, ok=%d" % (name, dt * 1e3, int((r.status == 0).sum())), flush=True)
        if envs is None:
            full = r.cpu()
        else:
            banded = r.cpu()
    same = sum(int(np.array_equal(banded.labels[i, :banded.out_len[i]], full.labels[i, :full.out_len[i]]))
               for i in range(B))
    print("consensus identical to the full-envelope search for %d of %d pairs" % (same, B))


if __name__ == "__main__":
    main()
