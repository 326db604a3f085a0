"""On-GPU timing probe for BASELINE config 5: python tools/probe_duplex.py [pairs] [T] [band] [mode]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import fast_ctc_decode_amd as fcd


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    def gen():
        x = torch.rand((B, T, 5), generator=g, device="cuda", dtype=torch.float32)
        return x / torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)
    x1, x2 = gen(), gen()
    i = np.arange(T)
    env = np.stack([np.maximum(0, i - w), np.minimum(T, i + w)], 1).astype(np.uint64)
    envs = torch.from_numpy(np.broadcast_to(env, (B, T, 2)).copy().view(np.int64)).cuda()
    torch.cuda.synchronize()
    for it in range(2):
        t0 = time.perf_counter()
        r = fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, logadd_mode=mode)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ms = r._handle.last_kernel_ms()
    print("duplex mode=%d pairs=%d T=%d band=+-%d: wall %.1f ms kernel %.1f ms  %.0f pairs/s  meanL=%.1f ok=%d"
          % (mode, B, T, w, dt * 1e3, ms, B / (ms / 1e3), float(r.out_len.float().mean()),
             int((r.status == 0).sum())), flush=True)


if __name__ == "__main__":
    main()
