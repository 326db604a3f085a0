"""Long randomised differential run (GPU vs oracle), beyond what the test suite samples:
    python tools/soak.py [n_seeds] [first_seed] [beam|beam_long|duplex_long|crf|crf_greedy|crf_duplex|viterbi|duplex|envelope]
Reuses the fuzz generators of tests/test_gpu_parity.py; prints one line per failing seed."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

import signal

import fast_ctc_decode_amd as fcd
import test_gpu_parity as tp

STATE = {"done": 0, "bad": 0, "which": ""}


def _on_term(signum, frame):  # `timeout` ends an open-ended soak: say how far it got
    try:  # (os.write: print() is not re-entrant and the signal may land inside one)
        os.write(1, ("soak %s (stopped by signal): %d seeds, %d failures\n"
                     % (STATE["which"], STATE["done"], STATE["bad"])).encode())
    finally:
        os._exit(1 if STATE["bad"] else 0)


signal.signal(signal.SIGTERM, _on_term)


def other(which, n, first):
    import test_gpu_duplex as td
    bad = 0
    t0 = time.time()
    STATE["which"] = which
    for seed in range(first, first + n):
        STATE["done"], STATE["bad"] = seed - first, bad
        try:
            if which == "crf":
                tp.crf_fuzz_seed(fcd, seed)
            elif which == "viterbi":
                tp.viterbi_fuzz_seed(fcd, seed)
            elif which == "beam_long":
                # long reads: FIFO refills, row reloads, jump-pointer traceback across many segments
                rng = np.random.default_rng(seed)
                N = int(rng.integers(3, 8))
                T = int(rng.integers(200, 3000))
                beam = int(rng.choice([1, 2, 5, 5, 8, 10, 12, 20]))
                x = tp.gen_batch(seed, int(rng.integers(1, 4)), T, N, peaky=bool(rng.integers(0, 2)))
                thr = float(rng.choice([0.0, 0.001, 0.1]))
                lengths = rng.integers(1, T + 1, size=x.shape[0]).astype(np.int64) if rng.integers(0, 2) else None
                for kernel in (0, 1, 2, 3, 4):
                    try:
                        tp.check_beam(fcd, x, beam, thr, True, lengths=lengths, kernel=kernel)
                    except RuntimeError as e:
                        assert kernel in (2, 3, 4) and " kernel: " in str(e), (seed, kernel, str(e))
            elif which == "duplex_long":
                rng = np.random.default_rng(seed)
                T1, T2 = int(rng.integers(100, 500)), int(rng.integers(100, 500))
                w = int(rng.integers(4, 70))
                x1, x2 = td.pairs(seed, 2, T1, T2, 5)
                i = np.arange(T1)
                c = (i * T2) // T1
                env = np.stack([np.maximum(0, c - w), np.minimum(T2, c + w + 1)], 1).astype(np.uint64)
                env[0, 0], env[-1, 1] = 0, T2
                env[1:, 0] = np.minimum(env[1:, 0], env[:-1, 1])
                envs = np.broadcast_to(env, (2, T1, 2)).copy()
                for mode in (td.LSE, td.MAX):
                    beam, thr = int(rng.choice([3, 5, 8])), float(rng.choice([0.0, 0.1]))
                    want = td.oracle_strings(x1, x2, "NACGT", envs, beam, thr, True, mode | td.CR)
                    got = td.gpu_strings(fcd, x1, x2, "NACGT", envs, beam, thr, True, mode)
                    assert got == want, (seed, T1, T2, w, beam, thr, mode)
            elif which == "crf_duplex":
                td.crf_duplex_fuzz_seed(fcd, seed, td.LSE)
                td.crf_duplex_fuzz_seed(fcd, seed, td.MAX)
            elif which == "crf_greedy":
                tp.crf_greedy_fuzz_seed(fcd, seed)
            elif which == "envelope":
                import test_gpu_envelope as te
                te.test_envelope_equals_model(fcd, seed)
            else:
                td.duplex_fuzz_seed(fcd, seed, td.LSE)
                td.duplex_fuzz_seed(fcd, seed, td.MAX)
        except AssertionError as e:
            bad += 1
            print("MISMATCH %s seed %d: %s" % (which, seed, str(e)[:300]), flush=True)
    print("soak %s: %d seeds, %d failures, %.1f s" % (which, n, bad, time.time() - t0), flush=True)
    return 1 if bad else 0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    which = sys.argv[3] if len(sys.argv) > 3 else "beam"
    if which != "beam":
        return other(which, n, first)
    bad = 0
    t0 = time.time()
    STATE["which"] = which
    for seed in range(first, first + n):
        STATE["done"], STATE["bad"] = seed - first, bad
        x, beam, thr, collapse, lengths = tp._fuzz_case(seed)
        for kernel in (0, 1, 2, 3, 4):
            try:
                tp.check_beam(fcd, x, beam, thr, collapse, lengths=lengths, kernel=kernel)
            except RuntimeError as e:
                if not (kernel in (2, 3, 4) and " kernel: " in str(e)):
                    bad += 1
                    print("seed %d kernel %d: %s" % (seed, kernel, e), flush=True)
            except AssertionError as e:
                bad += 1
                print("MISMATCH seed %d kernel %d beam %d thr %g collapse %s shape %s: %s"
                      % (seed, kernel, beam, thr, collapse, x.shape, str(e)[:200]), flush=True)
    print("soak: %d seeds x 5 kernel selections, %d failures, %.1f s" % (n, bad, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
