"""Quick on-GPU timing probe (not the bench contract): python tools/probe.py [B] [T] [beam] [kernel]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import fast_ctc_decode_amd as fcd


def gen(B, T, N, seed=1, device="cuda"):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.rand((B, T, N), generator=g, device=device, dtype=torch.float32)
    return x / torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    beam = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    kernel = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    x = gen(B, T, 5)
    torch.cuda.synchronize()
    for name, fn in (("viterbi", lambda: fcd.viterbi_search_batch_raw(x)),
                     ("beam%d/k%d" % (beam, kernel),
                      lambda: fcd.beam_search_batch_raw(x, beam, 0.1, kernel=kernel))):
        r = fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ms = r._handle.last_kernel_ms()
        L = float(r.out_len.float().mean())
        ok = int((r.status == 0).sum())
        print("%-12s B=%d T=%d  wall %.2f ms  kernel %.2f ms  %.0f reads/s  meanL=%.1f ok=%d"
              % (name, B, T, min(ts) * 1e3, ms, B / (ms / 1e3), L, ok), flush=True)


if __name__ == "__main__":
    main()
