"""Differential soak of the TIE-FLAGGED step (csrc/pdq178_reg.h, pdq178_wave.h) on the GPU:

    FCD_SOAK_SECONDS=240 python tools/tie_soak.py [first_seed] [n_seeds]

Posteriors quantised to a few levels tie at almost every step (tests/test_gpu_tieorder.py's generator); the shapes are
drawn so that a step has more than 20 candidates -- where sort_unstable_by's order of equal probabilities is the
quicksort's business -- on every kernel family that covers them: the register kernel at two reads and at one read per
wavefront (beams 6 .. 12: the list sorted in registers without ever going to memory), the lane kernel, the generic
one.  Every result is compared with the oracle under the same (default) order.  Prints cases / mismatches.
FCD_SOAK_EMU=1: the same run on the lockstep emulation of the kernels (tests/hipemu; CPU only, ~100x slower)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import fast_ctc_decode_amd as fcd
import test_gpu_parity as P
from beam_soak import budgeted


def case(seed):
    rng = np.random.default_rng(seed)
    wide = rng.random() < 0.25
    if wide:  # lane / generic kernels
        N = int(rng.integers(4, 9))
        beam = int(rng.integers(13, 65))
    else:  # the register kernel at one read per wavefront (and whatever else takes the shape)
        N = int(rng.integers(3, 8))
        beam = int(rng.integers(max(5, 21 // N + 1), 9 if N > 5 else 13))
    B = int(rng.integers(1, 9))
    T = int(rng.integers(20, 260))
    levels = int(rng.integers(2, 7))
    x = (rng.integers(0, levels, size=(B, T, N)) / float(levels)).astype(np.float32)
    if rng.random() < 0.5:
        x[:, :, 0] = np.maximum(x[:, :, 0], 1.0 / levels)
    if rng.random() < 0.3:  # a smooth stretch: ties come and go
        t0 = int(rng.integers(0, T))
        x[:, t0:t0 + 40] = rng.dirichlet(np.ones(N), size=(B, min(40, T - t0))).astype(np.float32)
    thr = float(rng.choice([0.0, 0.0, 0.05, 0.1, 1.0 / levels]))
    lengths = None
    if rng.random() < 0.3:
        lengths = rng.integers(0, T + 1, size=B).astype(np.int64)
    return x, beam, thr, bool(rng.random() < 0.8), lengths


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 9700000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9
    cases = bad = 0
    if os.environ.get("FCD_SOAK_EMU"):
        from emu_util import emulated_kernels
        ctx = emulated_kernels()
    else:
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        cases, bad = run(first, n)
    print("tie soak: seeds %d..%d, %d cases, %d mismatches" % (first, budgeted.last, cases, bad), flush=True)
    return 1 if bad else 0


def run(first, n):
    cases = bad = 0
    for seed in budgeted(first, n):
        x, beam, thr, collapse, lengths = case(seed)
        for kernel in (0, 1, 2, 3, 4):
            cases += 1
            try:
                P.check_beam(fcd, x, beam, thr, collapse, lengths=lengths, kernel=kernel)
            except RuntimeError as e:  # a forced kernel that does not cover the shape says so
                if not (kernel in (2, 3, 4) and " kernel: " in str(e)):
                    bad += 1
                    print("ERROR tie", seed, kernel, str(e)[:160], flush=True)
            except AssertionError as e:
                bad += 1
                print("MISMATCH tie", seed, kernel, x.shape, beam, thr, str(e)[:160], flush=True)
    return cases, bad


if __name__ == "__main__":
    sys.exit(main())
