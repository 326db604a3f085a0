"""Randomised differential soak of the duplex searches against the correctly-rounded oracle, on the GPU:

    python tools/duplex_soak.py [first_seed] [n_seeds]

tests/test_gpu_duplex.py's fuzz cases (random shapes, beams, thresholds, valid / invalid envelopes) plus the same cases
with special posteriors injected (exactly 1, exactly 0, above 1, NaN), plain and CRF, both log-add modes.
Prints the number of cases and of mismatches (0 expected)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import fast_ctc_decode_amd as fcd
import test_gpu_duplex as D



def budgeted(first, n):
    """seeds first .. first + n - 1, or as many as FCD_SOAK_SECONDS of wall clock allow (the summary line names the last one)"""
    import time
    budget = float(os.environ.get("FCD_SOAK_SECONDS", "0"))
    t0 = time.time()
    for seed in range(first, first + n):
        if budget and time.time() - t0 > budget:
            break
        budgeted.last = seed
        yield seed


budgeted.last = -1

def long_case(seed, mode):
    """Long reads inside a band (stale windows re-entering the beam, catch-up of several rows, discards, wobble)."""
    rng = np.random.default_rng(seed)
    N = 5
    B = 2
    T1 = int(rng.integers(300, 700))
    T2 = int(T1 * (0.9 + 0.2 * rng.random()))
    beam = int(rng.choice([3, 5, 8]))
    thr = float(rng.choice([0.0, 0.05, 0.1]))
    x1, x2 = D.pairs(seed, B, T1, T2, N)
    w = int(rng.integers(16, 65))
    env = D.band(T1, T2, w).astype(np.int64)
    if rng.integers(0, 2):  # a wobbly band: monotone bounds that move by 0..3 rows per step
        lo = np.minimum.accumulate(env[::-1, 0])[::-1]
        hi = np.maximum.accumulate(env[:, 1])
        jitter = rng.integers(0, 3, size=T1)
        hi = np.minimum(T2, np.maximum.accumulate(hi + jitter))
        env = np.stack([lo, hi], 1)
    envs = np.stack([env.astype(np.uint64)] * B)
    if rng.integers(0, 3) == 0:
        b, t, c = int(rng.integers(0, B)), int(rng.integers(0, T2)), int(rng.integers(0, N))
        x2[b, t, c] = [np.nan, 0.0, 1.0, 1.5][int(rng.integers(0, 4))]
    want = D.oracle_strings(x1, x2, "NACGT", envs, beam, thr, True, mode | D.CR)
    got = D.gpu_strings(fcd, x1, x2, "NACGT", envs, beam, thr, True, mode)
    return got == want


def main():
    if "--long" in sys.argv:
        sys.argv.remove("--long")
        first = int(sys.argv[1]) if len(sys.argv) > 1 else 900000
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
        cases = bad = 0
        for seed in budgeted(first, n):
            for mode in (D.LSE, D.MAX):
                cases += 1
                if not long_case(seed, mode):
                    bad += 1
                    print("MISMATCH long", seed, mode, flush=True)
        print("duplex soak (long reads): seeds %d..%d, %d cases, %d mismatches" % (first, budgeted.last, cases, bad))
        return 1 if bad else 0
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    cases = bad = 0
    for seed in budgeted(first, n):
        for mode in (D.LSE, D.MAX):
            for name, fn in (("fuzz", lambda: D.duplex_fuzz_seed(fcd, seed, mode)),
                             ("crf", lambda: D.crf_duplex_fuzz_seed(fcd, seed, mode))):
                cases += 1
                try:
                    fn()
                except AssertionError as e:
                    bad += 1
                    print("MISMATCH", name, seed, mode, str(e)[:200], flush=True)
            cases += 1
            if not D.special_values_case(fcd, seed, mode):
                bad += 1
                print("MISMATCH special", seed, mode, flush=True)
    print("duplex soak: seeds %d..%d, %d cases, %d mismatches" % (first, budgeted.last, cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
