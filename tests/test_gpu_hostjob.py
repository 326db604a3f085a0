"""The chunked host path (csrc/hostjob.hip): fcd_*_host on large batches and the fcd_*_host_begin /
fcd_job_next / fcd_job_end stream of result chunks must give exactly what the one-shot staging path gives --
chunk boundaries, ragged lengths, failing reads, every search, qualities, tie counters."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from test_gpu_parity import gen_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fcd():
    import fast_ctc_decode_amd as m
    return m


def _handle():
    from fast_ctc_decode_amd import _native as nat
    return nat.default_handle()


class pipeline:
    """with pipeline(lanes, chunk): the default handle's fcd_*_host calls take the chunked path whatever the size"""

    def __init__(self, lanes, chunk):
        self.args = (lanes, chunk, 0)

    def __enter__(self):
        _handle().set_host_pipeline(*self.args)

    def __exit__(self, *exc):
        _handle().set_host_pipeline(0, 0, -1)
        return False


class one_shot(pipeline):
    def __init__(self):
        self.args = (1, 0, -1)


def _same(a, b, what):
    assert np.array_equal(np.asarray(a.out_len), np.asarray(b.out_len)), what
    assert np.array_equal(np.asarray(a.status), np.asarray(b.status)), what
    for i in range(len(a.out_len)):
        n = int(a.out_len[i])
        assert np.array_equal(a.labels[i, :n], b.labels[i, :n]), (what, i)
        if a.path is not None:
            assert np.array_equal(a.path[i, :n], b.path[i, :n]), (what, i)
        if a.qual is not None:
            assert np.array_equal(a.qual[i, :n].view(np.uint32), b.qual[i, :n].view(np.uint32)), (what, i)
    if a.ambiguous is not None:
        assert np.array_equal(a.ambiguous, b.ambiguous), what


def _troubled_batch(seed, B, T, N):
    """ragged lengths incl. 0 and 1, a NaN row (IncomparableValues), an all-zero read (RanOutOfBeam)"""
    x = gen_batch(seed, B, T, N)
    lengths = np.random.default_rng(seed).integers(2, T + 1, B).astype(np.int64)
    lengths[1], lengths[min(6, B - 1)] = 0, 1
    x[3, 7, :] = np.nan
    x[B - 2] = 0.0
    lengths[3] = max(lengths[3], 9)
    lengths[B - 2] = max(lengths[B - 2], 5)
    return x, lengths


@pytest.mark.parametrize("lanes,chunk", [(3, 5), (2, 1), (4, 64), (2, 16)])
def test_host_pipeline_equals_one_shot(fcd, lanes, chunk):
    x, lengths = _troubled_batch(31, 17, 160, 5)
    for kernel, beam in ((0, 5), (1, 5), (4, 20)):
        with one_shot():
            a = fcd.beam_search_batch_raw(x, beam, 0.1, True, lengths=lengths, kernel=kernel, count_ambiguous=True)
        with pipeline(lanes, chunk):
            b = fcd.beam_search_batch_raw(x, beam, 0.1, True, lengths=lengths, kernel=kernel, count_ambiguous=True)
        _same(a, b, "beam kernel %d" % kernel)
        assert set(np.asarray(a.status).tolist()) >= {0, 1, 2}
    with one_shot():
        a = fcd.viterbi_search_batch_raw(x, True, lengths=lengths, qual=True)
    with pipeline(lanes, chunk):
        b = fcd.viterbi_search_batch_raw(x, True, lengths=lengths, qual=True)
    _same(a, b, "viterbi")
    # against the oracle too (the one-shot path is what every other parity test covers)
    for i in (0, 2, 5, 16):
        st, labels, path, _ = oracle.beam_search_raw(np.ascontiguousarray(x[i, :lengths[i]]), 5, 0.1, True)
        with pipeline(lanes, chunk):
            r = fcd.beam_search_batch_raw(x, 5, 0.1, True, lengths=lengths)
        assert int(r.status[i]) == st
        if st == 0:
            n = int(r.out_len[i])
            assert np.array_equal(r.labels[i, :n], labels) and np.array_equal(r.path[i, :n], path)


def test_host_pipeline_crf(fcd):
    rng = np.random.default_rng(5)
    B, T, S, N = 11, 120, 4, 5
    x = rng.random((B, T, S, N), dtype=np.float32)
    init = rng.random((B, S), dtype=np.float32)
    lengths = rng.integers(1, T + 1, B).astype(np.int64)
    with one_shot():
        a = fcd.crf_beam_search_batch_raw(x, init, 5, 0.0, lengths=lengths)
        g = fcd.crf_greedy_search_batch_raw(x, init, lengths=lengths, qual=True)
    with pipeline(3, 4):
        b = fcd.crf_beam_search_batch_raw(x, init, 5, 0.0, lengths=lengths)
        k = fcd.crf_greedy_search_batch_raw(x, init, lengths=lengths, qual=True)
    _same(a, b, "crf beam")
    _same(g, k, "crf greedy")


def test_host_pipeline_strided_views(fcd):
    """element strides that are not C-contiguous: every chunk uploads its own span"""
    big = gen_batch(77, 9, 2 * 90, 7)
    x = big[:, ::2, 1:6]  # (9, 90, 5) with stride_t = 14, stride_n = 1 and a column offset
    with one_shot():
        a = fcd.beam_search_batch_raw(x, 5, 0.05, True)
    with pipeline(2, 2):
        b = fcd.beam_search_batch_raw(x, 5, 0.05, True)
    _same(a, b, "strided")
    st, labels, path, _ = oracle.beam_search_raw(np.ascontiguousarray(x[8]), 5, 0.05, True)
    n = int(b.out_len[8])
    assert st == 0 and np.array_equal(b.labels[8, :n], labels) and np.array_equal(b.path[8, :n], path)


def _job_collect(h, lib, job, nat):
    out = []
    ch = nat.Chunk()
    while True:
        rc = lib.fcd_job_next(job, C.byref(ch))
        if rc == nat.JOB_DONE:
            break
        h.check(rc)
        n = int(ch.n_reads)
        lens = np.ctypeslib.as_array(C.cast(ch.out_len, C.POINTER(C.c_uint32)), (n,)).copy()
        stat = np.ctypeslib.as_array(C.cast(ch.status, C.POINTER(C.c_int32)), (n,)).copy()
        offs = np.ctypeslib.as_array(C.cast(ch.offsets, C.POINTER(C.c_uint64)), (n + 1,)).copy()
        total = int(offs[-1])
        labels = np.ctypeslib.as_array(C.cast(ch.labels, C.POINTER(C.c_uint8)), (max(total, 1),))[:total].copy()
        ptype = C.c_uint16 if ch.path_bytes == 2 else C.c_uint32
        path = np.ctypeslib.as_array(C.cast(ch.path, C.POINTER(ptype)), (max(total, 1),))[:total].copy() \
            if ch.path else None
        out.append((int(ch.read_begin), lens, stat, offs, labels, path))
    return out


def test_job_api_chunks_and_cancel(fcd):
    from fast_ctc_decode_amd import _native as nat
    from fast_ctc_decode_amd import api
    x, lengths = _troubled_batch(9, 23, 130, 5)
    with one_shot():
        ref = fcd.beam_search_batch_raw(x, 5, 0.1, True, lengths=lengths)
    h = _handle()
    lib = h.lib
    h.set_host_pipeline(3, 4, 0)
    try:
        b = api._host_batch(x, False, lengths)
        job = C.c_void_p()
        h.check(lib.fcd_beam_search_host_begin(h.ptr, C.byref(b), 5, 0.1, 1, 0, nat.JOB_PATH, C.byref(job)))
        chunk, lanes = C.c_int64(0), C.c_int(0)
        assert lib.fcd_job_chunks(job, C.byref(chunk), C.byref(lanes)) == 6 and chunk.value == 4 and lanes.value == 3
        # a second job on the same handle is refused while this one runs
        job2 = C.c_void_p()
        assert lib.fcd_beam_search_host_begin(h.ptr, C.byref(b), 5, 0.1, 1, 0, 0, C.byref(job2)) == nat.E_INVALID
        chunks = _job_collect(h, lib, job, nat)
        h.check(lib.fcd_job_end(job))
        assert [c[0] for c in chunks] == [0, 4, 8, 12, 16, 20]
        for begin, lens, stat, offs, labels, path in chunks:
            for i in range(len(lens)):
                r = begin + i
                assert lens[i] == ref.out_len[r] and stat[i] == ref.status[r]
                assert np.array_equal(labels[offs[i]:offs[i + 1]], ref.labels[r, :lens[i]])
                assert np.array_equal(path[offs[i]:offs[i + 1]], ref.path[r, :lens[i]])
        # abandon a job after its first chunk; the handle must be usable afterwards
        h.check(lib.fcd_beam_search_host_begin(h.ptr, C.byref(b), 5, 0.1, 1, 0, nat.JOB_PATH, C.byref(job)))
        ch = nat.Chunk()
        h.check(lib.fcd_job_next(job, C.byref(ch)))
        h.check(lib.fcd_job_end(job))
        again = fcd.beam_search_batch_raw(x, 5, 0.1, True, lengths=lengths)
        _same(ref, again, "after a cancelled job")
        # a job whose search cannot run reports the failure through fcd_job_next
        h.check(lib.fcd_beam_search_host_begin(h.ptr, C.byref(b), 5, 0.1, 1, nat.KERNEL_WAVE, nat.JOB_PATH,
                                               C.byref(job)))
        h.check(lib.fcd_job_end(job))
        bad = api._host_batch(gen_batch(1, 4, 40, 12), False)
        h.check(lib.fcd_beam_search_host_begin(h.ptr, C.byref(bad), 5, 0.01, 1, nat.KERNEL_WAVE, 0, C.byref(job)))
        assert lib.fcd_job_next(job, C.byref(ch)) == nat.E_UNSUPPORTED
        h.check(lib.fcd_job_end(job))
        # empty batch: no chunks
        empty = api._host_batch(np.zeros((0, 10, 5), np.float32), False)
        h.check(lib.fcd_beam_search_host_begin(h.ptr, C.byref(empty), 5, 0.1, 1, 0, 0, C.byref(job)))
        assert lib.fcd_job_next(job, C.byref(ch)) == nat.JOB_DONE
        h.check(lib.fcd_job_end(job))
    finally:
        h.set_host_pipeline(0, 0, -1)


def test_host_pipeline_default_thresholds(fcd):
    """BASELINE config-2 rows through fcd_beam_search_host at a size that takes the pipeline by default
    (>= 128 reads, >= 16 MB): identical to the device path on the same reads."""
    import torch
    x = gen_batch(123, 512, 4000, 5)
    r_host = fcd.beam_search_batch_raw(x, 5, 0.1, True)
    r_dev = fcd.beam_search_batch_raw(torch.from_numpy(x).cuda(), 5, 0.1, True).cpu()
    _same(r_host, r_dev, "host pipeline vs device path")
    assert (np.asarray(r_host.status) == 0).all()


def _compiled_layer():
    from fast_ctc_decode_amd import api
    return api._compiled()


@pytest.mark.parametrize("lanes,chunk", [(3, 2), (1, 0)])
def test_compiled_batch_functions_equal_per_read_calls(fcd, lanes, chunk):
    """The compiled module's *_batch functions (what fast_ctc_decode_amd's host batch API calls): element i is
    the per-read function's result for read i -- list / array / no paths, quality strings, multi-character
    labels, ragged arrays given as one padded array + lengths or as a list of per-read arrays."""
    cm = _compiled_layer()
    cm._set_host_pipeline(lanes, chunk, 0)
    try:
        x = gen_batch(3, 9, 120, 5)
        lengths = np.array([120, 0, 1, 50, 120, 77, 3, 119, 64])
        for paths in ("list", "array", None):
            res = fcd.beam_search_batch(x, "NACGT", 5, 0.1, lengths=lengths, paths=paths)
            assert len(res) == 9
            for i in range(9):
                ref = fcd.beam_search(x[i, :lengths[i]], "NACGT", 5, 0.1)
                s, p = res[i]
                assert s == ref[0]
                if paths == "list":
                    assert p == ref[1] and all(type(v) is int for v in p)
                elif paths == "array":
                    assert isinstance(p, np.ndarray) and p.dtype == np.uint32 and p.tolist() == ref[1]
                else:
                    assert p is None
        reads = [x[i, :lengths[i]] for i in range(9) if lengths[i] > 0]
        res = fcd.viterbi_search_batch(reads, "NACGT", qstring=True, qscale=1.3, qbias=0.5)
        assert res == [fcd.viterbi_search(r, "NACGT", qstring=True, qscale=1.3, qbias=0.5) for r in reads]
        alpha = ["", "Ab", "C", "éè", "T"]  # multi-character and non-ASCII labels
        assert fcd.beam_search_batch(x[:4], alpha, 5, 0.05) == [fcd.beam_search(x[i], alpha, 5, 0.05) for i in range(4)]
        rng = np.random.default_rng(11)
        x4 = rng.random((5, 60, 4, 5), dtype=np.float32)
        init = rng.random((5, 4), dtype=np.float32)
        assert fcd.crf_beam_search_batch(x4, init, alpha, 5, 0.0) == \
            [fcd.crf_beam_search(x4[i], init[i], alpha, 5, 0.0) for i in range(5)]
        assert fcd.crf_greedy_search_batch(x4, init, "NACGT", qstring=True) == \
            [fcd.crf_greedy_search(x4[i], init[i], "NACGT", qstring=True) for i in range(5)]
        # a failing read: RuntimeError naming the read, or None with raise_on_error=False
        bad = x.copy()
        bad[2, 5, :] = np.nan
        with pytest.raises(RuntimeError, match=r"read 2: Failed to compare values"):
            fcd.beam_search_batch(bad, "NACGT", 5, 0.1)
        res = cm.beam_search_batch(bad, "NACGT", 5, 0.1, raise_on_error=False)
        assert res[2] is None and res[3] == fcd.beam_search(bad[3], "NACGT", 5, 0.1)
        # argument errors are the per-read functions' (same checks, same order)
        with pytest.raises(ValueError, match="alphabet size 4 does not match"):
            fcd.beam_search_batch(x, "NACG", 5, 0.1)
        with pytest.raises(ValueError, match="beam_size cannot be 0"):
            fcd.beam_search_batch(x, "NACGT", 0, 0.1)
        with pytest.raises(TypeError):
            fcd.beam_search_batch(x.astype(np.float64), "NACGT", 5, 0.1)
        assert fcd.beam_search_batch(np.zeros((0, 10, 5), np.float32), "NACGT") == []
    finally:
        cm._set_host_pipeline(0, 0, -1)


def test_compiled_duplex_batch_functions_equal_per_read_calls(fcd):
    """The compiled module's duplex batch functions (and fast_ctc_decode_amd's, which call them on host inputs): element
    i is the per-read function's result for pair i -- both log-add modes, explicit and default envelopes, ragged
    pairs, a failing pair by exception or as None."""
    import test_gpu_duplex as D
    cm = _compiled_layer()
    rng = np.random.default_rng(5)
    B, T1, T2, N = 5, 40, 44, 5
    x1 = rng.random((B, T1, N), dtype=np.float32)
    x2 = rng.random((B, T2, N), dtype=np.float32)
    x1 /= np.linalg.norm(x1, axis=-1, keepdims=True)
    x2 /= np.linalg.norm(x2, axis=-1, keepdims=True)
    env1 = D.band(T1, T2, 12)
    envs = np.broadcast_to(env1, (B, T1, 2)).copy()
    for mode in ("logsumexp", "max"):
        want = [fcd.beam_search_duplex(x1[i], x2[i], "NACGT", env1, 5, 0.1, logadd_mode=mode) for i in range(B)]
        assert fcd.beam_search_duplex_batch(x1, x2, "NACGT", envs, 5, 0.1, logadd_mode=mode) == want
        assert cm.beam_search_duplex_batch(x1, x2, "NACGT", envs, 5, 0.1, logadd_mode=mode) == want
        want_full = [fcd.beam_search_duplex(x1[i], x2[i], "NACGT", None, 5, 0.1, logadd_mode=mode) for i in range(B)]
        assert fcd.beam_search_duplex_batch(x1, x2, "NACGT", None, 5, 0.1, logadd_mode=mode) == want_full
    # ragged second reads: the default envelope ends at each pair's own length
    l2 = np.array([44, 30, 44, 17, 40])
    got = cm.beam_search_duplex_batch(x1, x2, "NACGT", None, 5, 0.1, lengths_2=l2, logadd_mode="max")
    assert got == [fcd.beam_search_duplex(x1[i], x2[i, :l2[i]], "NACGT", None, 5, 0.1, logadd_mode="max") for i in range(B)]
    # a failing pair
    bad = envs.copy()
    bad[3, 7] = (30, 20)  # an envelope row whose bounds cross
    with pytest.raises(RuntimeError, match=r"pair 3: "):
        cm.beam_search_duplex_batch(x1, x2, "NACGT", bad, 5, 0.1)
    res = cm.beam_search_duplex_batch(x1, x2, "NACGT", bad, 5, 0.1, raise_on_error=False)
    assert res[3] is None and res[4] == fcd.beam_search_duplex(x1[4], x2[4], "NACGT", env1, 5, 0.1)
    with pytest.raises(ValueError, match="envelopes must have shape"):
        cm.beam_search_duplex_batch(x1, x2, "NACGT", envs[:, :-1], 5, 0.1)
    with pytest.raises(ValueError, match="alphabet size 4 does not match"):
        cm.beam_search_duplex_batch(x1, x2, "NACG", envs, 5, 0.1)
    # CRF pairs
    S = 4
    c1 = rng.random((B, T1, S, N), dtype=np.float32)
    c2 = rng.random((B, T2, S, N), dtype=np.float32)
    i1 = rng.random((B, S), dtype=np.float32)
    i2 = rng.random((B, S), dtype=np.float32)
    for mode in ("logsumexp", "max"):
        want = [fcd.crf_beam_search_duplex(c1[i], i1[i], c2[i], i2[i], "NACGT", env1, 5, 0.05, logadd_mode=mode)
                for i in range(B)]
        assert fcd.crf_beam_search_duplex_batch(c1, i1, c2, i2, "NACGT", envs, 5, 0.05, logadd_mode=mode) == want
        assert cm.crf_beam_search_duplex_batch(c1, i1, c2, i2, "NACGT", envs, 5, 0.05, logadd_mode=mode) == want


def test_list_paths_reference_counts(fcd):
    """list[int] paths share ONE int object per row index, their lists are filled by worker threads and the reference
    counts settled in bulk: the counts must come out exactly as if every entry had been stored with its own
    Py_INCREF -- the results equal the per-read calls' and dropping them returns the shared ints to their baseline."""
    import gc
    import sys
    cm = _compiled_layer()
    rng = np.random.default_rng(1)
    x = rng.random((12, 700, 5), dtype=np.float32)
    x /= np.linalg.norm(x, axis=-1, keepdims=True)
    res = cm.beam_search_batch(x, "NACGT", 5, 0.1, paths="list")
    assert res == [fcd.beam_search(x[i], "NACGT", 5, 0.1) for i in range(12)]
    entries = [v for _, p in res for v in p if v > 256]        # (ints above the interpreter's small-int cache)
    some = entries[len(entries) // 2]
    uses = sum(1 for _, p in res for v in p if v is some)
    assert uses >= 1
    with_results = sys.getrefcount(some)
    n_in_entries = sum(1 for v in entries if v is some)
    del entries, res
    gc.collect()
    assert with_results - sys.getrefcount(some) == uses + n_in_entries


def test_time_major_host_views_through_the_batch_functions(fcd):
    """A time-major (T, B, N) score array handed over as `scores.transpose(1, 0, 2)` -- float32 and float16 -- goes
    through the compiled batch functions without a copy on the Python side and decodes as its contiguous copy does."""
    cm = _compiled_layer()
    rng = np.random.default_rng(3)
    T, B, N = 90, 11, 5
    xt = rng.random((T, B, N), dtype=np.float32)
    xt /= np.linalg.norm(xt, axis=-1, keepdims=True)
    for dt in (np.float32, np.float16):
        view = xt.astype(dt).transpose(1, 0, 2)
        assert not view.flags["C_CONTIGUOUS"]
        want = cm.beam_search_batch(np.ascontiguousarray(view), "NACGT", 5, 0.1)
        assert cm.beam_search_batch(view, "NACGT", 5, 0.1) == want
        assert fcd.beam_search_batch(view, "NACGT", 5, 0.1) == want
        assert cm.viterbi_search_batch(view, "NACGT") == cm.viterbi_search_batch(np.ascontiguousarray(view), "NACGT")


@pytest.mark.parametrize("lanes,chunk", [(3, 3), (1, 0)])
def test_list_of_ragged_reads_without_a_padded_copy(fcd, lanes, chunk):
    """A Python list of per-read (T_r, N) arrays -- how the reference's callers hold their reads (src/lib.rs:325,352) --
    goes through fcd_*_host_ptrs_begin: each chunk is gathered from the reads' own memory by its lane, nothing is
    padded on the Python side.  Element i == the per-read call on read i: ragged lengths incl. an empty read, a
    non-contiguous read (copied once, by numpy), several chunks per lane, list / array / no paths, viterbi qualities."""
    cm = _compiled_layer()
    cm._set_host_pipeline(lanes, chunk, 0)
    try:
        x = gen_batch(21, 17, 150, 5)
        rows = [150, 1, 77, 0, 149, 64, 65, 3, 150, 10, 99, 128, 127, 150, 2, 31, 150]
        reads = [np.ascontiguousarray(x[i, :rows[i]]) for i in range(17)]
        reads[4] = x[4, :rows[4] * 1][::1]                       # a view
        reads[6] = np.asfortranarray(x[6, :rows[6]])              # not C-contiguous
        for paths in ("list", "array", None):
            res = cm.beam_search_batch(reads, "NACGT", 5, 0.1, paths=paths)
            assert len(res) == 17
            for i, r in enumerate(reads):
                want = fcd.beam_search(np.ascontiguousarray(r), "NACGT", 5, 0.1) if rows[i] > 0 else ("", [])
                s, p = res[i]
                assert s == want[0], i
                if paths == "list":
                    assert p == want[1]
                elif paths == "array":
                    assert p.tolist() == want[1]
        # big-endian float32 has float32's type number and other bytes: it must not be read as native floats (ADVICE
        # r4) -- it is not a float32 array in the reference's sense either (PyO3's &PyArray2<f32>): TypeError
        with pytest.raises(TypeError):
            cm.beam_search_batch([r.astype(">f4") for r in reads], "NACGT", 5, 0.1)
        nonempty = [r for r in reads if r.shape[0] > 0]
        assert cm.viterbi_search_batch(nonempty, "NACGT", qstring=True) == \
            [fcd.viterbi_search(np.ascontiguousarray(r), "NACGT", qstring=True) for r in nonempty]
        with pytest.raises(TypeError):
            cm.beam_search_batch([reads[0], x[1, :, :4]], "NACGT", 5, 0.1)   # inner shapes differ
        with pytest.raises(ValueError, match="alphabet size 4 does not match"):
            cm.beam_search_batch(reads, "NACG", 5, 0.1)
    finally:
        cm._set_host_pipeline(0, 0, -1)
