"""Single-call latency of the drop-in per-read functions (host numpy in, Python objects out)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import fast_ctc_decode_amd as fcd


def rows(T, N, seed):
    rng = np.random.default_rng(seed)
    x = rng.random((T, N), dtype=np.float32)
    return x / np.linalg.norm(x, ord=2, axis=1, keepdims=True)


def timeit(fn, n=20):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def main():
    for T in (100, 1000, 4000):
        x = rows(T, 5, T)
        print("T=%5d  viterbi_search %.3f ms   beam_search(5, 0.1) %.3f ms   beam_search(32, 0.1) %.3f ms"
              % (T, timeit(lambda: fcd.viterbi_search(x, "NACGT")),
                 timeit(lambda: fcd.beam_search(x, "NACGT", 5, 0.1)),
                 timeit(lambda: fcd.beam_search(x, "NACGT", 32, 0.1))), flush=True)
    try:
        import fast_ctc_decode as ext  # compiled module (pybind11)
        x = rows(4000, 5, 1)
        print("compiled module: T=4000 beam_search %.3f ms  viterbi %.3f ms"
              % (timeit(lambda: ext.beam_search(x, "NACGT", 5, 0.1)), timeit(lambda: ext.viterbi_search(x, "NACGT"))))
    except Exception as e:  # noqa: BLE001
        print("compiled module not importable:", e)


if __name__ == "__main__":
    main()
