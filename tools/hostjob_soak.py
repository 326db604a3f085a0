"""Randomised soak of the host batch path (csrc/hostjob.hip + csrc/pymodule.cpp): the compiled module's *_batch functions
under random pipeline settings (lanes, chunk sizes), ragged and empty reads, failing reads, list / array / no paths,
quality strings, against the per-read functions -- the same kernels behind both, so this is about the host logic:
chunk boundaries, packed downloads, object building.  Runs on the GPU, or on the CPU under the emulator:

    python tools/hostjob_soak.py [first_seed] [n_seeds]            (FCD_TEST_EMU=1 for the emulator)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np



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

def run(fcd, first, n):
    from fast_ctc_decode_amd import api
    cm = api._compiled()
    cases = bad = 0
    for seed in budgeted(first, n):
        rng = np.random.default_rng(seed)
        B, T, N = int(rng.integers(1, 40)), int(rng.integers(1, 120)), int(rng.integers(2, 7))
        lanes, chunk = int(rng.integers(1, 5)), int(rng.integers(1, 9))
        cm._set_host_pipeline(lanes, chunk, 0)
        x = rng.random((B, T, N), dtype=np.float32)
        x /= np.linalg.norm(x, axis=-1, keepdims=True)
        lengths = rng.integers(0, T + 1, size=B) if rng.integers(0, 2) else None
        for _ in range(int(rng.integers(0, 3))):
            x[int(rng.integers(0, B)), int(rng.integers(0, T)), :] = np.nan   # a read that fails to compare
        beam = int(rng.choice([1, 3, 5, 9, 20]))
        thr = float(rng.choice([0.0, 0.05, 0.1]))
        alpha = "NACGTUV"[:N]
        paths = [None, "list", "array"][int(rng.integers(0, 3))]
        cases += 1
        try:
            got = cm.beam_search_batch(x, alpha, beam, thr, True, lengths, paths, 0, False)
            for i in range(B):
                Ti = T if lengths is None else int(lengths[i])
                try:
                    want = fcd.beam_search(x[i, :Ti], alpha, beam, thr) if Ti > 0 else ("", [])
                except RuntimeError:
                    want = None
                if want is None:
                    assert got[i] is None, (i, "should have failed")
                    continue
                assert got[i] is not None and got[i][0] == want[0], (i, "sequence")
                if paths == "list":
                    assert got[i][1] == want[1]
                elif paths == "array":
                    assert got[i][1].tolist() == want[1]
                else:
                    assert got[i][1] is None
            # viterbi with quality strings on the same batch
            q = cm.viterbi_search_batch(x, alpha, True, 1.0, 0.0, True, lengths, "list", False)
            for i in range(B):
                Ti = T if lengths is None else int(lengths[i])
                if Ti == 0:
                    continue
                assert q[i] == fcd.viterbi_search(x[i, :Ti], alpha, qstring=True), (i, "viterbi")
        except AssertionError as e:
            bad += 1
            print("MISMATCH", seed, B, T, N, lanes, chunk, paths, str(e)[:160], flush=True)
    cm._set_host_pipeline(0, 0, -1)
    print("host batch soak: seeds %d..%d, %d cases, %d mismatches" % (first, budgeted.last, cases, bad))
    return 1 if bad else 0


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 700000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    if os.environ.get("FCD_TEST_EMU"):
        from emu_util import emulated_kernels
        import fast_ctc_decode_amd as fcd
        with emulated_kernels():
            return run(fcd, first, n)
    import fast_ctc_decode_amd as fcd
    return run(fcd, first, n)


if __name__ == "__main__":
    sys.exit(main())
