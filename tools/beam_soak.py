"""Randomised differential soak of the 1-D searches (beam_search on every kernel selection, crf_beam_search,
viterbi_search, crf_greedy_search) against the oracle, on the GPU, with special posteriors injected:

    python tools/beam_soak.py [first_seed] [n_seeds]

tests/test_gpu_parity.py's fuzz draws (random shapes, beams, thresholds, ragged lengths, quantised ties) with a few
entries overwritten by NaN, +inf, values above 1, exact zeros and negative numbers.  Prints cases / mismatches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import fast_ctc_decode_amd as fcd
import test_gpu_parity as P



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

def inject(rng, x):
    n = int(rng.integers(0, 5))
    for _ in range(n):
        idx = tuple(int(rng.integers(0, s)) for s in x.shape)
        kind = int(rng.integers(0, 5))
        x[idx] = [np.nan, np.inf, 1.0 + float(rng.random()), 0.0, -0.25][kind]
    return x


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    cases = bad = 0
    from fast_ctc_decode_amd import _native as nat
    h = nat.default_handle()
    for seed in budgeted(first, n):
        x, beam, thr, collapse, lengths = P._fuzz_case(seed)
        rng = np.random.default_rng(seed + 7)
        x = inject(rng, x)
        # r06: half the seeds under a small workspace limit with first-pass slabs of a sixth of the worst case (the wide-beam
        # kernel then takes its slabs from the device-side pool, a handful of them, and most reads go through the retry
        # pass; the other kernels run in chunks), a third with the handle's calls on internal streams (fcd_set_overlap)
        knobs = np.random.default_rng(seed + 29)
        limited = knobs.random() < 0.5
        h.set_workspace_limit(int(knobs.integers(1, 9)) << 20 if limited else 0)
        h.check(h.lib.fcd_debug_set_first_pass_divisor(h.ptr, 6 if limited else 0))
        h.set_overlap(int(knobs.integers(2, 6)) if knobs.random() < 0.33 else 0)
        for kernel in (0, 1, 2, 3, 4):
            cases += 1
            try:
                P.check_beam(fcd, x, beam, thr, collapse, lengths=lengths, kernel=kernel)
            except RuntimeError as e:  # a forced kernel that does not cover the shape says so
                if not (kernel in (2, 3, 4) and " kernel: " in str(e)):
                    bad += 1
                    print("ERROR beam", seed, kernel, str(e)[:160], flush=True)
            except AssertionError as e:
                bad += 1
                print("MISMATCH beam", seed, kernel, beam, thr, collapse, str(e)[:160], flush=True)
        # the same reads held as float16 / bfloat16-representable values: the kernels convert on load (fcd_batch.dtype)
        xh = x.astype(np.float16)
        cases += 1
        try:
            r = fcd.beam_search_batch_raw(xh, beam, thr, collapse, lengths=lengths)
            up = xh.astype(np.float32)
            for i in range(x.shape[0]):
                xi = up[i] if lengths is None else up[i, :lengths[i]]
                st, labels, path, _ = P.oracle.beam_search_raw(np.ascontiguousarray(xi), beam, thr, collapse)
                assert int(r.status[i]) == st
                if st == 0:
                    m = int(r.out_len[i])
                    assert m == len(labels) and np.array_equal(r.labels[i, :m], labels) and np.array_equal(r.path[i, :m], path)
        except AssertionError as e:
            bad += 1
            print("MISMATCH float16", seed, beam, thr, str(e)[:160], flush=True)
        # wide beams (the lane-per-entry kernel, one or two reads per wavefront) and wide alphabets (the LDS kernel)
        rng2 = np.random.default_rng(seed + 13)
        Nw = int(rng2.integers(2, 13))
        beam_w = int(rng2.choice([17, 24, 32, 33, 48, 64]))
        Tw, Bw = int(rng2.integers(1, 260)), int(rng2.integers(1, 5))
        style = int(rng2.integers(0, 3))
        if style == 0:
            xw = P.reference_style_rows(rng2, Bw * Tw, Nw).reshape(Bw, Tw, Nw)
        elif style == 1:
            xw = (rng2.integers(0, 4, size=(Bw, Tw, Nw)) / 4.0).astype(np.float32)
        else:
            xw = np.ldexp(1.0, -rng2.integers(0, 5, size=(Bw, Tw, Nw))).astype(np.float32)
        xw = inject(rng2, np.ascontiguousarray(xw, np.float32))
        thr_w = float(rng2.choice([0.0, 0.02, 0.1]))
        for kernel in (0, 1, 4):
            cases += 1
            try:
                P.check_beam(fcd, xw, beam_w, thr_w, bool(rng2.integers(0, 2)), kernel=kernel)
            except RuntimeError as e:
                if not (kernel == 4 and " kernel: " in str(e)):
                    bad += 1
                    print("ERROR wide", seed, kernel, str(e)[:160], flush=True)
            except AssertionError as e:
                bad += 1
                print("MISMATCH wide", seed, kernel, Nw, beam_w, thr_w, str(e)[:160], flush=True)
        # viterbi (+ quality values) on the same reads
        cases += 1
        try:
            r = fcd.viterbi_search_batch_raw(x, collapse, lengths=lengths, qual=True)
            for i in range(x.shape[0]):
                Ti = x.shape[1] if lengths is None else int(lengths[i])
                if Ti == 0:
                    assert int(r.out_len[i]) == 0
                    continue
                labels, path, quals = P.oracle.viterbi_search_raw(np.ascontiguousarray(x[i, :Ti]), collapse)
                m = int(r.out_len[i])
                assert m == len(labels) and np.array_equal(r.labels[i, :m], labels) and np.array_equal(r.path[i, :m], path)
                got = [P.oracle.lib.fcdo_phred(float(q), 1.0, 0.0) for q in r.qual[i, :m]]
                assert [ord(c) for c in got] == list(quals)
        except AssertionError as e:
            bad += 1
            print("MISMATCH viterbi", seed, str(e)[:160], flush=True)
        # time-major storage, (T, B, N) seen as a batch by its strides: the time-major viterbi kernel (groups of 8 reads)
        rng3 = np.random.default_rng(seed + 29)
        Bt, Tt, Nt = int(rng3.integers(8, 41)), int(rng3.integers(1, 200)), int(rng3.integers(2, 9))
        xt = inject(rng3, (rng3.integers(0, 5, size=(Tt, Bt, Nt)) / 4.0).astype(np.float32))
        if rng3.integers(0, 2):
            xt = xt.astype(np.float16)
        view = xt.transpose(1, 0, 2)
        lens_t = rng3.integers(0, Tt + 1, size=Bt).astype(np.int64) if rng3.integers(0, 2) else None
        want_q = bool(rng3.integers(0, 2))
        cases += 1
        try:
            r = fcd.viterbi_search_batch_raw(view, collapse, lengths=lens_t, qual=want_q)
            up = np.ascontiguousarray(view).astype(np.float32)
            for i in range(Bt):
                Ti = Tt if lens_t is None else int(lens_t[i])
                if Ti == 0:
                    assert int(r.out_len[i]) == 0
                    continue
                labels, path, quals = P.oracle.viterbi_search_raw(np.ascontiguousarray(up[i, :Ti]), collapse)
                m = int(r.out_len[i])
                assert m == len(labels) and np.array_equal(r.labels[i, :m], labels) and np.array_equal(r.path[i, :m], path)
                if want_q:
                    got = [P.oracle.lib.fcdo_phred(float(q), 1.0, 0.0) for q in r.qual[i, :m]]
                    assert [ord(c) for c in got] == list(quals)
        except AssertionError as e:
            bad += 1
            print("MISMATCH viterbi time-major", seed, Bt, Tt, Nt, str(e)[:160], flush=True)
        # CRF searches: S states x N symbols out of the same generator
        S = int(rng.choice([4, 16]))
        T, B = int(rng.integers(1, 90)), int(rng.integers(1, 4))
        xc = inject(rng, rng.random((B, T, S, 5), dtype=np.float32))
        init = rng.random((B, S), dtype=np.float32)
        for kernel in (0, 1):
            cases += 1
            try:
                r = fcd.crf_beam_search_batch_raw(xc, init, beam, thr, kernel=kernel)
                for i in range(B):
                    st, labels, path, _ = P.oracle.crf_beam_search_ambiguous(xc[i], init[i], beam, thr)
                    assert int(r.status[i]) == st, (i, int(r.status[i]), st)
                    if st == 0:
                        m = int(r.out_len[i])
                        assert m == len(labels) and np.array_equal(r.labels[i, :m], labels) and \
                            np.array_equal(r.path[i, :m], path)
            except AssertionError as e:
                bad += 1
                print("MISMATCH crf", seed, kernel, S, beam, thr, str(e)[:160], flush=True)
        # crf_greedy_search, read-major and time-major storage of the same reads (B >= 4: the time-major kernel)
        rng4 = np.random.default_rng(seed + 41)
        Sg = int(rng4.choice([2, 4, 8]))
        Ng = int(rng4.integers(2, 5)) if Sg == 8 else int(rng4.integers(2, 7))
        Bg, Tg = int(rng4.integers(4, 14)), int(rng4.integers(1, 150))
        xg = inject(rng4, (rng4.integers(0, 6, size=(Bg, Tg, Sg, Ng)) / 5.0).astype(np.float32) + np.float32(0.01))
        ig = rng4.random((Bg, Sg), dtype=np.float32)
        lens_g = rng4.integers(0, Tg + 1, size=Bg).astype(np.int64) if rng4.integers(0, 2) else None
        tm = np.ascontiguousarray(xg.transpose(1, 0, 2, 3)).transpose(1, 0, 2, 3)
        for name, arr in (("read-major", xg), ("time-major", tm)):
            cases += 1
            try:
                r = fcd.crf_greedy_search_batch_raw(arr, ig, lengths=lens_g)
                for i in range(Bg):
                    Ti = Tg if lens_g is None else int(lens_g[i])
                    if Ti == 0:
                        continue
                    try:
                        seq, path = P.oracle.crf_greedy_search(np.ascontiguousarray(xg[i, :Ti]), ig[i], "NACGTUV"[:Ng])
                    except RuntimeError:
                        assert int(r.status[i]) != 0, (i, "should have failed")
                        continue
                    m = int(r.out_len[i])
                    assert int(r.status[i]) == 0 and "".join("NACGTUV"[l] for l in r.labels[i, :m]) == seq and \
                        np.array_equal(r.path[i, :m], path), (i, name)
            except AssertionError as e:
                bad += 1
                print("MISMATCH crf_greedy", name, seed, Bg, Tg, Sg, Ng, str(e)[:160], flush=True)
    print("beam soak: seeds %d..%d, %d cases, %d mismatches" % (first, budgeted.last, cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
