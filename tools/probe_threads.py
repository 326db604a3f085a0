"""Per-read callers on many threads (the reference's calling pattern, src/lib.rs:199/:353 release the GIL for
it): reads/s of the compiled drop-in module with and without the coalescing front door.

    python tools/probe_threads.py [T] [calls_per_thread] [threads ...]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import fast_ctc_decode as ext


def rows(T, seed):
    rng = np.random.default_rng(seed)
    x = rng.random((T, 5), dtype=np.float32)
    return x / np.linalg.norm(x, ord=2, axis=1, keepdims=True)


def run(n_threads, calls, reads, fn):
    def work(tid):
        for j in range(calls):
            fn(reads[(tid * calls + j) % len(reads)])
    ts = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return n_threads * calls / (time.perf_counter() - t0)


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    threads = [int(a) for a in sys.argv[3:]] or [1, 4, 16, 64]
    reads = [rows(T, s) for s in range(64)]
    for name, fn in (("beam_search(5, 0.1)", lambda x: ext.beam_search(x, "NACGT", 5, 0.1)),
                     ("viterbi_search", lambda x: ext.viterbi_search(x, "NACGT"))):
        for n in threads:
            ext.set_coalescing(0)
            run(n, 2, reads, fn)
            plain = run(n, calls, reads, fn)
            ext.set_coalescing(256, 0)
            run(n, 2, reads, fn)
            co = run(n, calls, reads, fn)
            st = ext.coalescing_stats()
            ext.set_coalescing(0)
            print("T=%d %-20s %3d threads: %9.0f reads/s per-read launches, %9.0f reads/s coalesced "
                  "(%d calls in %d launches, largest batch %d)"
                  % (T, name, n, plain, co, st["calls"], st["launches"], st["largest_batch"]), flush=True)


def main_pairs():
    """python tools/probe_threads.py pairs [T] [calls_per_thread] [threads ...]: per-PAIR callers of beam_search_duplex
    (src/lib.rs:401-488: one pair per call), banded envelope of +-64 rows as in BASELINE config 5"""
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    threads = [int(a) for a in sys.argv[4:]] or [1, 16, 64]
    i = np.arange(T)
    env = np.stack([np.maximum(i - 64, 0), np.minimum(i + 64, T)], 1).astype(np.uint64)
    pairs = [(rows(T, 2 * s), rows(T, 2 * s + 1)) for s in range(64)]
    fn = lambda p: ext.beam_search_duplex(p[0], p[1], "NACGT", env, 5, 0.1)
    for n in threads:
        ext.set_coalescing(0)
        run(n, 1, pairs, fn)
        plain = run(n, calls, pairs, fn)
        ext.set_coalescing(256, 0)
        run(n, 1, pairs, fn)
        co = run(n, calls, pairs, fn)
        st = ext.coalescing_stats()
        ext.set_coalescing(0)
        print("T=%d beam_search_duplex(5, 0.1, band 64) %3d threads: %7.1f pairs/s per-pair launches, %7.1f pairs/s coalesced "
              "(%d calls in %d launches, largest batch %d)"
              % (T, n, plain, co, st["calls"], st["launches"], st["largest_batch"]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pairs":
        main_pairs()
    else:
        main()
