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


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    cases = bad = 0
    for seed in range(first, first + n):
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
    print("duplex soak: seeds %d..%d, %d cases, %d mismatches" % (first, first + n - 1, cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
