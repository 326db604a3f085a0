"""Workload for rocprofv3 runs: a few launches of each hot kernel at the BASELINE shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import fast_ctc_decode_amd as fcd


def gen(B, T, N, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    x = torch.rand((B, T, N), generator=g, device="cuda", dtype=torch.float32)
    return x / torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = 3
    if which in ("all", "beam"):
        x = gen(4096, 4000, 5, 1)
        for _ in range(reps):
            r = fcd.beam_search_batch_raw(x, 5, 0.1, True)
        torch.cuda.synchronize()
        print("beam ok", int((r.status == 0).sum()), "mean L", float(r.out_len.float().mean()))
    if which in ("all", "viterbi"):
        x = gen(16384, 4000, 5, 2)
        for _ in range(reps):
            r = fcd.viterbi_search_batch_raw(x)
        torch.cuda.synchronize()
        print("viterbi mean L", float(r.out_len.float().mean()))
        xh = x.to(torch.float16)   # half-precision posteriors read directly (fcd_batch.dtype)
        for _ in range(reps):
            r = fcd.viterbi_search_batch_raw(xh)
        torch.cuda.synchronize()
        print("viterbi f16 mean L", float(r.out_len.float().mean()))
        del xh
    if which in ("all", "beam32"):
        x = gen(8192, 4000, 5, 3)
        for _ in range(2):
            r = fcd.beam_search_batch_raw(x, 32, 0.1, True)
        torch.cuda.synchronize()
        print("beam32 ok", int((r.status == 0).sum()))
        del x
    if which in ("all", "crf"):
        g = torch.Generator(device="cuda")
        g.manual_seed(3)
        x = torch.rand((4096, 4000, 4, 5), generator=g, device="cuda")
        x = x / torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)
        init = torch.zeros((4096, 4), device="cuda")
        init[torch.arange(4096), torch.arange(4096) % 4] = 1.0
        for _ in range(reps):
            r = fcd.crf_beam_search_batch_raw(x, init, 5, 0.0)
        torch.cuda.synchronize()
        print("crf ok", int((r.status == 0).sum()))
        del x
    if which in ("all", "duplex"):
        import numpy as np
        B, T, w = 1024, 2000, 64
        x1, x2 = gen(B, T, 5, 4), gen(B, T, 5, 5)
        i = np.arange(T)
        env = np.stack([np.maximum(0, i - w), np.minimum(T, i + w)], 1).astype(np.uint64)
        envs = torch.from_numpy(np.broadcast_to(env, (B, T, 2)).copy().view(np.int64)).cuda()
        for mode in (0, 1):
            r = fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, logadd_mode=mode)
            torch.cuda.synchronize()
            print("duplex mode", mode, "ok", int((r.status == 0).sum()))


if __name__ == "__main__":
    main()
