"""The coalescing front door (include/fcd.h fcd_coalescer_*, csrc/coalesce.hip): concurrent per-read calls from
many threads must return exactly what the per-read calls return, while sharing batched launches.

CPU: through the lockstep emulation of the kernels (tests/hipemu); `-m gpu`: on the MI355X, both host layers."""
import threading

import numpy as np
import pytest

from kat_cases import reference_style_rows
from oracle import oracle

ALPHA = "NACGT"


def _reads(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        T = int(rng.integers(1, 160))
        x = reference_style_rows(rng, T, 5).reshape(T, 5)
        if i % 5 == 3:
            x = np.asfortranarray(x)       # strided input
        out.append(x)
    out[2][out[2].shape[0] // 2] = np.nan  # IncomparableValues for the beam search of read 2 (if it has >= 2 rows)
    return out


def _expected_beam(x, beam, thr):
    st, labels, path, _ = oracle.beam_search_raw(np.ascontiguousarray(x), beam, thr, True)
    if st != 0:
        return ("error", st)
    return ("".join(ALPHA[l] for l in labels), [int(p) for p in path])


def _expected_viterbi(x):
    labels, path = oracle.viterbi_search_raw(np.ascontiguousarray(x), True)[:2]
    return ("".join(ALPHA[l] for l in labels), [int(p) for p in path])


def _hammer(mod, reads, n_threads, params):
    """Every thread decodes its share of the reads with per-read calls; -> {(read, kind): result}."""
    results, errors = {}, []

    def work(tid):
        try:
            for i in range(tid, len(reads), n_threads):
                x = reads[i]
                beam, thr = params[i % len(params)]
                try:
                    results[(i, "beam")] = tuple(mod.beam_search(x, ALPHA, beam, thr))
                except RuntimeError as e:
                    results[(i, "beam")] = ("error", str(e))
                results[(i, "viterbi")] = tuple(mod.viterbi_search(x, ALPHA))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    return results


def _check(mod, reads, n_threads, max_wait_us):
    from fast_ctc_decode_amd import _native as nat
    params = [(5, 0.1), (5, 0.0), (3, 0.05), (32, 0.1)]
    mod.set_coalescing(16, max_wait_us)
    try:
        got = _hammer(mod, reads, n_threads, params)
        stats = mod.coalescing_stats()
    finally:
        mod.set_coalescing(0)
    assert mod.coalescing_stats() is None
    for i, x in enumerate(reads):
        beam, thr = params[i % len(params)]
        want = _expected_beam(x, beam, thr)
        if want[0] == "error":
            assert got[(i, "beam")] == ("error", nat.status_string(want[1])), (i, got[(i, "beam")])
        else:
            assert got[(i, "beam")] == (want[0], want[1]), i
        assert got[(i, "viterbi")] == _expected_viterbi(x), i
    assert stats["calls"] == 2 * len(reads)
    return stats


def _crf_reads(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        T = int(rng.integers(1, 120))
        S = 4 if i % 3 else 16
        x = rng.random((T, S, 5), dtype=np.float32)
        if i % 4 == 1:
            x = np.ascontiguousarray(x.transpose(1, 0, 2)).transpose(1, 0, 2)  # strided (T, S, N) view
        init = rng.random(S, dtype=np.float32)
        out.append((x, init))
    return out


def _check_crf(mod, reads, n_threads, max_wait_us):
    """crf_beam_search / crf_greedy_search through the coalescer == the same calls without it"""
    def call(i):
        x, init = reads[i]
        beam, thr = [(5, 0.0), (3, 0.01), (32, 0.0)][i % 3]
        try:
            a = tuple(mod.crf_beam_search(x, init, ALPHA, beam, thr))
        except RuntimeError as e:
            a = ("error", str(e))
        return a, tuple(mod.crf_greedy_search(x, init, ALPHA, qstring=(i % 2 == 0)))

    want = [call(i) for i in range(len(reads))]
    got, errors = {}, []

    def work(tid):
        try:
            for i in range(tid, len(reads), n_threads):
                got[i] = call(i)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    mod.set_coalescing(16, max_wait_us)
    try:
        threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        stats = mod.coalescing_stats()
    finally:
        mod.set_coalescing(0)
    assert not errors, errors
    for i in range(len(reads)):
        assert got[i] == want[i], i
    assert stats["calls"] == 2 * len(reads)
    return stats


def _pairs(n, seed):
    """pairs of reads with a banded envelope each (ragged lengths; a few with the default envelope), and CRF pairs"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        T1, T2 = int(rng.integers(8, 60)), int(rng.integers(8, 60))
        x1 = reference_style_rows(rng, T1, 5).reshape(T1, 5)
        x2 = reference_style_rows(rng, T2, 5).reshape(T2, 5)
        if i % 4 == 1:
            x2 = np.asfortranarray(x2)
        if i % 3 == 0:
            env = None
        else:
            c = np.arange(T1) * T2 // T1
            lo = np.maximum(c - 6, 0)
            hi = np.minimum(c + 7, T2)
            lo[0], hi[-1] = 0, T2
            env = np.stack([lo, hi], 1).astype(np.uint64)
        out.append((x1, x2, env))
    return out


def _crf_pairs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        T1, T2, S = int(rng.integers(6, 40)), int(rng.integers(6, 40)), 4
        out.append((rng.random((T1, S, 5), dtype=np.float32), rng.random(S, dtype=np.float32),
                    rng.random((T2, S, 5), dtype=np.float32), rng.random(S, dtype=np.float32)))
    return out


def _check_pairs(mod, pairs, crf_pairs, n_threads, max_wait_us):
    """beam_search_duplex / crf_beam_search_duplex through the coalescer == the same calls without it"""
    def call(i):
        if i < len(pairs):
            x1, x2, env = pairs[i]
            beam, thr = [(5, 0.1), (3, 0.0)][i % 2]
            try:
                return mod.beam_search_duplex(x1, x2, ALPHA, env, beam, thr)
            except RuntimeError as e:
                return ("error", str(e))
        x1, i1, x2, i2 = crf_pairs[i - len(pairs)]
        try:
            return mod.crf_beam_search_duplex(x1, i1, x2, i2, ALPHA, None, 5, 0.0)
        except RuntimeError as e:
            return ("error", str(e))

    n = len(pairs) + len(crf_pairs)
    want = [call(i) for i in range(n)]
    assert sum(isinstance(w, str) and len(w) > 0 for w in want) > n // 2  # real sequences, not a wall of errors
    got, errors = {}, []

    def work(tid):
        try:
            for i in range(tid, n, n_threads):
                got[i] = call(i)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    mod.set_coalescing(16, max_wait_us)
    try:
        threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        stats = mod.coalescing_stats()
    finally:
        mod.set_coalescing(0)
    assert not errors, errors
    for i in range(n):
        assert got[i] == want[i], i
    assert stats["calls"] == n
    return stats


def test_coalescer_emulated():
    import fast_ctc_decode_amd as fcd
    from emu_util import emulated_kernels
    with emulated_kernels():
        stats = _check(fcd, _reads(24, 5), 6, 20000)
        crf_stats = _check_crf(fcd, _crf_reads(12, 7), 4, 20000)
    # six threads start together and the leader waits 20 ms for company: launches are shared
    assert stats["launches"] < stats["calls"] and stats["largest_batch"] >= 2, stats
    assert crf_stats["launches"] < crf_stats["calls"], crf_stats


def test_coalescer_pair_searches_emulated():
    """r05: the per-PAIR searches (src/lib.rs:401-578) through the same door, both host layers"""
    import fast_ctc_decode_amd as fcd
    from emu_util import emu_compiled_module, emulated_kernels
    with emulated_kernels():
        stats = _check_pairs(fcd, _pairs(10, 11), _crf_pairs(4, 12), 5, 20000)
        assert stats["launches"] < stats["calls"] and stats["largest_batch"] >= 2, stats
        stats = _check_pairs(emu_compiled_module(), _pairs(8, 13), _crf_pairs(4, 14), 4, 20000)
        assert stats["launches"] < stats["calls"] and stats["largest_batch"] >= 2, stats


@pytest.mark.gpu
@pytest.mark.parametrize("layer", ["mirror", "compiled"])
def test_coalescer_gpu(layer):
    if layer == "mirror":
        import fast_ctc_decode_amd as mod
    else:
        import fast_ctc_decode as mod
    reads = _reads(96, 6)
    stats = _check(mod, reads, 16, 0)         # no timer: batches form from whatever arrives during a launch
    assert stats["launches"] <= stats["calls"], stats
    stats = _check(mod, reads, 16, 2000)
    assert stats["launches"] < stats["calls"] and stats["largest_batch"] >= 2, stats
    stats = _check_crf(mod, _crf_reads(60, 8), 12, 2000)
    assert stats["launches"] < stats["calls"] and stats["largest_batch"] >= 2, stats
    stats = _check_pairs(mod, _pairs(40, 9), _crf_pairs(12, 10), 12, 2000)
    assert stats["launches"] < stats["calls"] and stats["largest_batch"] >= 2, stats


@pytest.mark.gpu
def test_coalescer_rejects_what_it_cannot_batch():
    import ctypes as C

    from fast_ctc_decode_amd import _native as nat
    co = nat.Coalescer(0, 8, 0)
    try:
        x = np.zeros((2, 4, 5), np.float32)
        b = nat.Batch(x.ctypes.data, 2, 4, 1, 5, 20, 5, 0, 1, None)    # two reads: not a per-read call
        lab, ln, st = np.zeros(4, np.uint8), np.zeros(1, np.uint32), np.zeros(1, np.int32)
        res = nat.Result(lab.ctypes.data, None, None, ln.ctypes.data, st.ctypes.data, 4, None)
        with pytest.raises(nat.NativeError, match="exactly one"):
            co.check(co.lib.fcd_coalescer_beam_search(co.ptr, C.byref(b), 5, 0.1, 1, C.byref(res)))
    finally:
        co.close()
