"""End-to-end cost of the batch API around the kernel: python tools/probe_e2e.py [reads]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import fast_ctc_decode_amd as fcd


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rng = np.random.default_rng(1)
    x = rng.random((B * 4000, 5), dtype=np.float32)
    x /= np.linalg.norm(x, ord=2, axis=1, keepdims=True)
    x = x.reshape(B, 4000, 5)
    xd = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()

    def t(fn, n=3):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return min(ts) * 1e3, r

    ms_raw_dev, r = t(lambda: fcd.beam_search_batch_raw(xd, 5, 0.1))
    ms_cpu, rc = t(lambda: r.cpu())
    ms_seq, seqs = t(lambda: rc.sequences("NACGT"))
    ms_seq_a, _ = t(lambda: rc.sequences("NACGT", paths="array"))
    ms_host, _ = t(lambda: fcd.beam_search_batch(x, "NACGT", 5, 0.1), n=2)
    ms_host_a, _ = t(lambda: fcd.beam_search_batch(x, "NACGT", 5, 0.1, paths="array"), n=2)
    print("B=%d: device raw %.1f ms | results to host %.1f ms | build (str, path) objects %.1f ms | "
          "beam_search_batch(host numpy -> python objects) %.1f ms" % (B, ms_raw_dev, ms_cpu, ms_seq, ms_host))
    print("with paths='array': build objects %.1f ms | beam_search_batch end to end %.1f ms" % (ms_seq_a, ms_host_a))
    print("mean len", np.mean([len(s) for s, _ in seqs]))


if __name__ == "__main__":
    main()
