"""Differential parity: HIP kernels (through the C ABI) vs the CPU oracle on seeded inputs.
Bar: bit-exact (labels, path, status) -- this is integer/index output."""
import numpy as np
import pytest

from kat_cases import reference_style_rows
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fcd():
    import fast_ctc_decode_amd as m
    return m


def gen_batch(seed, B, T, N, peaky=False):
    rng = np.random.default_rng(seed)
    if peaky:
        z = rng.normal(size=(B, T, N)).astype(np.float32) * 4.0
        z[..., 0] += 2.0
        e = np.exp(z - z.max(-1, keepdims=True))
        return (e / e.sum(-1, keepdims=True)).astype(np.float32)
    return reference_style_rows(rng, B * T, N).reshape(B, T, N)


def check_beam(fcd, x, beam, thr, collapse=True, lengths=None, kernel=0):
    r = fcd.beam_search_batch_raw(x, beam, thr, collapse, lengths=lengths, kernel=kernel)
    B = x.shape[0]
    for i in range(B):
        xi = x[i] if lengths is None else x[i, :lengths[i]]
        st, labels, path, _ = oracle.beam_search_raw(np.ascontiguousarray(xi), beam, thr, collapse)
        assert int(r.status[i]) == st, (i, int(r.status[i]), st)
        if st == 0:
            n = int(r.out_len[i])
            assert n == len(labels), (i, n, len(labels))
            np.testing.assert_array_equal(r.labels[i, :n], labels)
            np.testing.assert_array_equal(r.path[i, :n], path)


@pytest.mark.parametrize("N", [3, 5, 12])
@pytest.mark.parametrize("beam", [1, 5, 32])
def test_beam_generic_random(fcd, N, beam):
    x = gen_batch(100 + N + beam, 6, 300, N)
    thr = 0.1 if N <= 5 else 0.05
    check_beam(fcd, x, beam, thr, kernel=fcd.KERNEL_GENERIC)


KERNELS = [1, 2, 3]  # generic (LDS), wave (two reads per wavefront where possible), wave1


def test_largest_beam_of_the_lds_kernel(fcd):
    """README: beams up to ~295 at N = 5 on the LDS-resident kernel.  The quicksort's node-ordered list and scratch
    (FCD_TIE_PDQ178, more than 20 candidates) need LDS of their own, so that order stops earlier (247) and says so;
    the stable order keeps the whole 64 KiB (the space used to be reserved whatever the order)."""
    from tie_util import tie_order
    x = gen_batch(77, 2, 12, 5)

    def largest(lo, hi):  # largest beam_size in [lo, hi) that is accepted
        while hi - lo > 1:
            mid = (lo + hi) // 2
            try:
                fcd.beam_search_batch_raw(x, mid, 0.0, True, kernel=fcd.KERNEL_GENERIC)
                lo = mid
            except RuntimeError as e:
                assert "LDS" in str(e)
                hi = mid
        return lo

    with tie_order(fcd, "stable"):
        assert largest(64, 600) == 295
        check_beam(fcd, x, 295, 0.0, kernel=fcd.KERNEL_GENERIC)
    with tie_order(fcd, "pdq178"):
        assert largest(64, 600) == 247
        check_beam(fcd, x, 247, 0.0, kernel=fcd.KERNEL_GENERIC)
        with pytest.raises(RuntimeError, match="FCD_TIE_STABLE"):
            fcd.beam_search_batch_raw(x, 280, 0.0, True, kernel=fcd.KERNEL_GENERIC)


@pytest.mark.parametrize("kernel", [2, 3])
@pytest.mark.parametrize("N", [3, 4, 5, 6, 7])
@pytest.mark.parametrize("beam", [1, 2, 5, 8])
def test_beam_wave_random(fcd, N, beam, kernel):
    x = gen_batch(200 + N * 10 + beam, 7, 400, N)
    check_beam(fcd, x, beam, 0.1 if N <= 5 else 0.05, kernel=kernel)


@pytest.mark.parametrize("kernel", [0, 2, 3])
@pytest.mark.parametrize("N", [3, 4, 5])
@pytest.mark.parametrize("beam", [9, 10, 12])
def test_beam_wave_wide(fcd, N, beam, kernel):
    """beam 9..12: the twelve-groups-of-five-lanes instantiation of the wave kernel."""
    x = gen_batch(900 + N * 10 + beam, 6, 500, N)
    check_beam(fcd, x, beam, 0.1, kernel=kernel)
    check_beam(fcd, x[:2], beam, 0.0, collapse=False, kernel=kernel)
    lengths = np.array([500, 1, 0, 65, 64, 499], np.int64)
    check_beam(fcd, x, beam, 0.05, lengths=lengths, kernel=kernel)


@pytest.mark.parametrize("N", [2, 3, 5, 8])
@pytest.mark.parametrize("beam", [1, 5, 13, 32, 64])
def test_beam_lane_random(fcd, N, beam):
    """One beam entry per lane (beam_size <= 64, N <= 8): seeded batches, thr 0 / 0.1, collapse on/off,
    ragged lengths, exact ties."""
    x = gen_batch(1300 + N * 10 + beam, 5, 400, N)
    thr = 0.1 if N <= 5 else 0.05
    check_beam(fcd, x, beam, thr, kernel=fcd.KERNEL_LANE)
    check_beam(fcd, x[:2], beam, 0.0, collapse=False, kernel=fcd.KERNEL_LANE)
    lengths = np.array([400, 1, 0, 65, 399], np.int64)
    check_beam(fcd, x, beam, thr / 2, lengths=lengths, kernel=fcd.KERNEL_LANE)
    rng = np.random.default_rng(beam * 100 + N)
    q = (rng.integers(0, 4, size=(3, 150, N)) / 4.0).astype(np.float32)   # exact ties: the all-pairs fallback
    check_beam(fcd, q, beam, 0.0, kernel=fcd.KERNEL_LANE)


def check_ambiguous(fcd, x, beam, thr, kernels, collapse=True):
    """fcd_result.ambiguous == the oracle's count of unpinned-tie steps, and the
    instrumented kernels return exactly what the oracle (and hence the timed kernels) return."""
    want = [oracle.beam_search_ambiguous(x[i], beam, thr, collapse) for i in range(len(x))]
    for k in kernels:
        r = fcd.beam_search_batch_raw(x, beam, thr, collapse, kernel=k, count_ambiguous=True)
        for i, (st, labels, path, n_amb) in enumerate(want):
            n = int(r.out_len[i])
            assert int(r.status[i]) == st, (k, i)
            if st == 0:
                np.testing.assert_array_equal(r.labels[i, :n], labels)
                np.testing.assert_array_equal(r.path[i, :n], path)
            assert tuple(int(v) for v in r.ambiguous[i]) == n_amb, (k, i, r.ambiguous[i], n_amb)
    return sum(w[3][0] for w in want), sum(w[3][1] for w in want)


def test_ambiguity_counter(fcd):
    """The tie instrument on every kernel family: quantised posteriors (exact ties at almost every step
    above 20 candidates) must give the oracle's non-zero counts; reference-style rows must give 0."""
    rng = np.random.default_rng(77)
    q = (rng.integers(0, 4, size=(5, 150, 5)) / 4.0).astype(np.float32)
    assert min(check_ambiguous(fcd, q, 5, 0.0, (0, 1, 2, 3, 4))) > 0
    assert min(check_ambiguous(fcd, q, 12, 0.0, (0, 1, 2, 4))) > 0
    assert min(check_ambiguous(fcd, q, 32, 0.0, (0, 1, 4), collapse=False)) > 0
    q8 = (rng.integers(0, 3, size=(3, 100, 8)) / 4.0).astype(np.float32)
    assert min(check_ambiguous(fcd, q8, 8, 0.0, (1, 4))) > 0
    assert min(check_ambiguous(fcd, q8, 64, 0.0, (1, 4))) > 0
    q4 = (rng.integers(0, 4, size=(3, 100, 4)) / 4.0).astype(np.float32)   # 5 x 4 = 20 candidates: never > 20
    n0, n1 = check_ambiguous(fcd, q4, 5, 0.0, (1, 2, 3, 4))
    assert n0 == 0 and n1 > 0
    x = gen_batch(78, 4, 300, 5)
    assert check_ambiguous(fcd, x, 5, 0.1, (0, 1, 2, 3, 4)) == (0, 0)
    assert check_ambiguous(fcd, x, 32, 0.1, (0, 1, 4)) == (0, 0)
    x[1, 100] = np.nan   # a read that fails mid-way reports the count up to the failing step
    check_ambiguous(fcd, x, 5, 0.1, (1, 2, 3, 4))


def test_lane_two_pass_retry(fcd):
    """Wide beams size the first-pass tree slabs below the worst case (capi.hip): reads that outgrow their slab
    are stopped and decoded again in worst-case slabs, a few per retry round.  A small workspace limit and a
    large first-pass divisor make ten of eleven short reads take that path, three rounds deep; results must be
    the oracle's, ragged lengths and a failing read included."""
    from fast_ctc_decode_amd import _native as nat
    rng = np.random.default_rng(3)
    B, T = 11, 300
    x = rng.random((B, T, 5), dtype=np.float32)
    x /= x.sum(-1, keepdims=True)
    x[3] = 0.2          # uniform rows: exact ties everywhere
    x[7, 150] = np.nan  # IncomparableValues half way
    lengths = np.array([T, T, 1, T, 0, T, 299, T, T, 64, T], np.int64)
    h = nat.default_handle()
    h.set_workspace_limit(2 << 20)
    h.check(h.lib.fcd_debug_set_first_pass_divisor(h.ptr, 6))
    try:
        for beam, thr in ((32, 0.0), (20, 0.1), (64, 0.0)):
            check_beam(fcd, x, beam, thr, lengths=lengths, kernel=fcd.KERNEL_LANE)
            check_beam(fcd, x, beam, thr, lengths=lengths, kernel=fcd.KERNEL_AUTO)
        # default sizing (half the worst case, adaptive): at threshold 0 every extension creates a node, every
        # full-length read overflows, and the handle sizes the NEXT job for the worst case -- same results
        h.check(h.lib.fcd_debug_set_first_pass_divisor(h.ptr, 0))
        for _ in range(2):
            check_beam(fcd, x, 32, 0.0, lengths=lengths, kernel=fcd.KERNEL_LANE)
    finally:
        h.set_workspace_limit(0)
        h.check(h.lib.fcd_debug_set_first_pass_divisor(h.ptr, 0))


def test_lane_overlapping_calls(fcd):
    """fcd_set_overlap (include/fcd.h): wide-beam calls go round-robin to internal streams and share ONE pool of tree slabs
    handed out on the device (slab_pool.h) -- here far fewer slabs than reads, so wavefronts wait for one another's.
    Five calls in flight on three streams must each deliver what the same call delivers in stream order; two calls
    that write the SAME result arrays must end with the second call's results."""
    import ctypes as C
    import torch
    from fast_ctc_decode_amd import _native as nat
    xs = [gen_batch(900 + i, 9, 200 + 16 * i, 5) for i in range(5)]
    h = nat.default_handle()
    h.set_workspace_limit(6 << 20)  # (the shortest batch fits it whole and takes the plain path: ordered behind the others too)
    h.check(h.lib.fcd_debug_set_first_pass_divisor(h.ptr, 6))
    on_gpu = torch.cuda.is_available()
    try:
        serial = [fcd.beam_search_batch_raw(x, 32, 0.05, True, kernel=fcd.KERNEL_LANE) for x in xs]
        for x, r in zip(xs[:2], serial[:2]):  # (and those are the oracle's)
            st, labels, path, _ = oracle.beam_search_raw(np.ascontiguousarray(x[0]), 32, 0.05, True)
            assert st == 0 and int(r.out_len[0]) == len(labels)
            np.testing.assert_array_equal(r.path[0, :len(labels)], path)
        h.set_overlap(3)
        if on_gpu:
            xt = [torch.from_numpy(x).cuda() for x in xs]
            rs = [fcd.beam_search_batch_raw(t, 32, 0.05, True, kernel=fcd.KERNEL_LANE) for t in xt]
            outs = [r.cpu() for r in rs]
        else:  # (the emulator: host arrays in, the staged call joins before it copies back)
            outs = [fcd.beam_search_batch_raw(x, 32, 0.05, True, kernel=fcd.KERNEL_LANE) for x in xs]
        for a, b in zip(outs, serial):
            np.testing.assert_array_equal(a.status, b.status)
            np.testing.assert_array_equal(a.out_len, b.out_len)
            for i in range(len(a.out_len)):
                n = int(a.out_len[i])
                np.testing.assert_array_equal(a.labels[i, :n], b.labels[i, :n])
                np.testing.assert_array_equal(a.path[i, :n], b.path[i, :n])
        if on_gpu:
            # the same result arrays twice: the second call is ordered behind the first although it sits on another stream
            B, w = xs[0].shape[0], max(x.shape[1] for x in xs)
            labels = torch.zeros((B, w), dtype=torch.uint8, device="cuda")
            path = torch.zeros((B, w), dtype=torch.int32, device="cuda")
            out_len = torch.zeros(B, dtype=torch.int32, device="cuda")
            status = torch.zeros(B, dtype=torch.int32, device="cuda")
            res = nat.Result(labels.data_ptr(), path.data_ptr(), None, out_len.data_ptr(), status.data_ptr(), w, None)
            h.set_stream(torch.cuda.current_stream().cuda_stream)
            for order in ((4, 1), (4, 3), (4, 4, 0), (3, 4, 2)):  # (the longest reads first: their call would finish last)
                for k in order:
                    t = xt[k]
                    st_ = t.stride()
                    b = nat.Batch(t.data_ptr(), t.shape[0], t.shape[1], 1, 5, st_[0], st_[1], 0, st_[2], None, nat.DTYPE_F32)
                    h.check(h.lib.fcd_beam_search_dev(h.ptr, C.byref(b), 32, 0.05, 1, fcd.KERNEL_LANE, C.byref(res)))
                h.overlap_join()
                torch.cuda.synchronize()
                want = serial[order[-1]]
                np.testing.assert_array_equal(out_len.cpu().numpy(), want.out_len)
                for i in range(B):
                    n = int(want.out_len[i])
                    np.testing.assert_array_equal(path[i, :n].cpu().numpy(), want.path[i, :n])
    finally:
        h.set_overlap(0)
        h.set_workspace_limit(0)
        h.check(h.lib.fcd_debug_set_first_pass_divisor(h.ptr, 0))


def test_module_level_overlap_switch(fcd):
    """fast_ctc_decode_amd.set_overlap / overlap_join (the thread's handle): results read through BatchResult.cpu() --
    which joins by itself -- and through tensors used directly after overlap_join() are those of stream order."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("device tensors only")
    xs = [torch.from_numpy(gen_batch(980 + i, 6, 180 + 20 * i, 5)).cuda() for i in range(4)]
    want = [fcd.beam_search_batch_raw(x, 5, 0.1, True).cpu() for x in xs]
    fcd.set_overlap(3)
    try:
        rs = [fcd.beam_search_batch_raw(x, 5, 0.1, True) for x in xs]
        fcd.overlap_join()                       # torch's current stream waits: the tensors themselves are good now
        lens = [r.out_len.clone() for r in rs]
        torch.cuda.synchronize()
        for ln, r, w in zip(lens, rs, want):
            np.testing.assert_array_equal(ln.cpu().numpy(), w.out_len)
            c = r.cpu()
            for i in range(len(w.out_len)):
                n = int(w.out_len[i])
                np.testing.assert_array_equal(c.path[i, :n], w.path[i, :n])
        seqs = fcd.beam_search_batch(xs[0], "NACGT", 5, 0.1)   # (the string-returning entry point joins too)
        assert [len(s) for s, _ in seqs] == [int(v) for v in want[0].out_len]
    finally:
        fcd.set_overlap(0)


@pytest.mark.parametrize("beam", [32, 40])
def test_lane_slab_pool_under_contention(fcd, beam):
    """The device-side slab pool (csrc/slab_pool.h) with far fewer slabs than wavefronts: a workspace limit leaves a
    handful of (pairs of) slabs for several hundred reads, so most wavefronts of the launch wait for a slab another one
    hands back -- and the retry pass (first-pass slabs of a sixth of the worst case: most reads overflow) queues for two or
    three worst-case slabs.  Same results as with a slab per read."""
    import torch
    from fast_ctc_decode_amd import _native as nat
    n = 640 if torch.cuda.is_available() else 48
    x = gen_batch(4100 + beam, n, 96, 5)
    lengths = np.full(n, 96, np.int64)
    lengths[::7] = np.arange(len(lengths[::7])) % 97
    want = fcd.beam_search_batch_raw(x, beam, 0.02, True, lengths=lengths, kernel=fcd.KERNEL_LANE)  # (fits: a slab per read)
    st, labels, path, _ = oracle.beam_search_raw(np.ascontiguousarray(x[1, :lengths[1]]), beam, 0.02, True)
    assert st == 0 and int(want.out_len[1]) == len(labels)
    np.testing.assert_array_equal(want.path[1, :len(labels)], path)
    h = nat.default_handle()
    h.set_workspace_limit(3 << 20)
    h.check(h.lib.fcd_debug_set_first_pass_divisor(h.ptr, 6))
    try:
        for _ in range(2):  # (the second call finds the pool as the first one left it)
            got = fcd.beam_search_batch_raw(x, beam, 0.02, True, lengths=lengths, kernel=fcd.KERNEL_LANE)
            np.testing.assert_array_equal(got.status, want.status)
            np.testing.assert_array_equal(got.out_len, want.out_len)
            for i in range(n):
                k = int(want.out_len[i])
                np.testing.assert_array_equal(got.labels[i, :k], want.labels[i, :k])
                np.testing.assert_array_equal(got.path[i, :k], want.path[i, :k])
    finally:
        h.set_workspace_limit(0)
        h.check(h.lib.fcd_debug_set_first_pass_divisor(h.ptr, 0))


@pytest.mark.parametrize("kernel,beam", [(0, 5), (1, 7), (3, 8), (4, 16)])
def test_overlapping_calls_every_kernel(fcd, kernel, beam):
    """fcd_set_overlap with the kernels that keep one tree slab per read: every internal stream has a region of the
    workspace to itself.  Six calls of different shapes on three streams (the workspace grows on the way: calls in flight
    are waited for first) deliver what they deliver in stream order."""
    import torch
    from fast_ctc_decode_amd import _native as nat
    xs = [gen_batch(950 + i, 5 + 3 * (i % 3), 150 + 40 * i, 5) for i in range(6)]
    h = nat.default_handle()
    on_gpu = torch.cuda.is_available()
    serial = [fcd.beam_search_batch_raw(x, beam, 0.05, True, kernel=kernel) for x in xs]
    st, labels, path, _ = oracle.beam_search_raw(np.ascontiguousarray(xs[5][0]), beam, 0.05, True)
    assert st == 0 and int(serial[5].out_len[0]) == len(labels)
    np.testing.assert_array_equal(serial[5].path[0, :len(labels)], path)
    h.set_overlap(3)
    try:
        if on_gpu:
            xt = [torch.from_numpy(x).cuda() for x in xs]
            outs = [fcd.beam_search_batch_raw(t, beam, 0.05, True, kernel=kernel) for t in xt]
            outs = [r.cpu() for r in outs]
        else:
            outs = [fcd.beam_search_batch_raw(x, beam, 0.05, True, kernel=kernel) for x in xs]
        for a, b in zip(outs, serial):
            np.testing.assert_array_equal(a.status, b.status)
            np.testing.assert_array_equal(a.out_len, b.out_len)
            for i in range(len(a.out_len)):
                n = int(a.out_len[i])
                np.testing.assert_array_equal(a.labels[i, :n], b.labels[i, :n])
                np.testing.assert_array_equal(a.path[i, :n], b.path[i, :n])
        # another entry point in between: it takes the workspace from its start, behind everything in flight
        if on_gpu:
            ra = fcd.beam_search_batch_raw(xt[5], beam, 0.05, True, kernel=kernel)
            rv = fcd.viterbi_search_batch_raw(xt[0], True)
            rb = fcd.beam_search_batch_raw(xt[4], beam, 0.05, True, kernel=kernel)
            for r, want in ((ra, serial[5]), (rb, serial[4])):
                c = r.cpu()
                np.testing.assert_array_equal(c.out_len, want.out_len)
                np.testing.assert_array_equal(c.path[0, :int(c.out_len[0])], want.path[0, :int(c.out_len[0])])
            assert int(rv.cpu().out_len[0]) > 0
    finally:
        h.set_overlap(0)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("collapse", [True, False])
def test_beam_thr0(fcd, collapse, kernel):
    x = gen_batch(7, 4, 200, 5)
    check_beam(fcd, x, 5, 0.0, collapse, kernel=kernel)


@pytest.mark.parametrize("kernel", KERNELS)
def test_beam_peaky(fcd, kernel):
    x = gen_batch(8, 8, 500, 5, peaky=True)
    check_beam(fcd, x, 5, 0.001, kernel=kernel)
    check_beam(fcd, x, 5, 0.1, kernel=kernel)  # many reads run out of beam: status parity


@pytest.mark.parametrize("kernel", KERNELS)
def test_beam_ragged(fcd, kernel):
    x = gen_batch(9, 7, 257, 5)
    lengths = np.array([257, 1, 0, 100, 64, 65, 128], np.int64)
    check_beam(fcd, x, 5, 0.1, lengths=lengths, kernel=kernel)


@pytest.mark.parametrize("kernel", KERNELS)
def test_beam_nan_and_zero_rows(fcd, kernel):
    x = gen_batch(10, 5, 50, 5)
    x[0, 20] = np.nan           # NaN row -> IncomparableValues
    x[1, 10:] = 0.0             # all-zero rows with thr 0.1 -> RanOutOfBeam
    x[2, 5, 0] = np.nan         # NaN blank only
    x[3, 7, 1:] = np.nan        # NaN labels only
    x[4, :, :] = np.nan         # K16: everything NaN
    check_beam(fcd, x, 5, 0.1, kernel=kernel)
    check_beam(fcd, x, 5, 0.0, kernel=kernel)
    check_beam(fcd, x, 1, 0.0, kernel=kernel)  # a lone NaN candidate is never compared


@pytest.mark.parametrize("kernel", [0, 1, 2, 3, 4])
def test_beam_special_posteriors(fcd, kernel):
    """search.rs:191,201 accept any f32: +inf, values above 1, negative numbers, exact zeros and scattered NaNs among
    ordinary rows (tools/beam_soak.py's injections, in the driver-run suite since r06) -- every kernel selection, narrow
    and wide beams, both thresholds; a forced kernel that does not cover a shape says so."""
    def inject(rng, x, n):
        for _ in range(n):
            idx = tuple(int(rng.integers(0, s_)) for s_ in x.shape)
            x[idx] = [np.nan, np.inf, 1.0 + float(rng.random()), 0.0, -0.25][int(rng.integers(0, 5))]
        return x
    rng = np.random.default_rng(9100 + kernel)
    cases = []
    for N, beam, T in ((5, 5, 120), (5, 3, 60), (4, 8, 90), (5, 32, 80), (7, 12, 70), (3, 64, 50)):
        x = reference_style_rows(rng, 6 * T, N).reshape(6, T, N)
        for b in range(6):  # read 0 stays ordinary; the others get 1 .. 5 special entries, one kind each at least once
            inject(rng, x[b], b)
        x[1, T // 2, 1] = np.inf
        x[2, T // 3, 0] = -0.25
        x[3, T // 4, 2] = 1.75
        x[4, T // 5, N - 1] = np.nan
        cases.append((np.ascontiguousarray(x, np.float32), beam))
    ran = 0
    for x, beam in cases:
        for thr in (0.0, 0.1):
            try:
                check_beam(fcd, x, beam, thr, kernel=kernel)
                ran += 1
            except RuntimeError as e:
                assert kernel in (2, 3, 4) and " kernel: " in str(e), str(e)
    assert ran >= 4, ran


@pytest.mark.parametrize("kernel", KERNELS)
def test_beam_ties_and_zeros(fcd, kernel):
    """Exact ties (equal probabilities, zeros) must resolve by ascending node index."""
    rng = np.random.default_rng(77)
    x = rng.integers(0, 3, size=(6, 300, 5)).astype(np.float32) * 0.25  # values in {0, .25, .5}
    x[:, :, 0] = np.maximum(x[:, :, 0], 0.25)
    check_beam(fcd, x, 5, 0.0, kernel=kernel)
    check_beam(fcd, x, 5, 0.1, kernel=kernel)
    check_beam(fcd, x, 3, 0.1, False, kernel=kernel)


@pytest.mark.parametrize("kernel", KERNELS)
def test_beam_denormals(fcd, kernel):
    """f32 subnormals must not be flushed (the CPU reference keeps them)."""
    x = gen_batch(12, 4, 200, 5) * np.float32(1e-38)
    check_beam(fcd, x, 5, 0.0, kernel=kernel)
    x2 = gen_batch(13, 4, 200, 5)
    x2[:, ::3, :] *= np.float32(1e-30)
    check_beam(fcd, x2, 5, 0.0, kernel=kernel)


def test_beam_strided_view(fcd):
    big = gen_batch(11, 3, 120, 8)
    x = big[:, ::2, 1:6]  # non-contiguous view, like a zero-copy ndarray view (src/lib.rs:352)
    r = fcd.beam_search_batch_raw(x, 5, 0.1)
    for i in range(3):
        st, labels, path, _ = oracle.beam_search_raw(x[i], 5, 0.1)
        n = int(r.out_len[i])
        assert int(r.status[i]) == st
        np.testing.assert_array_equal(r.labels[i, :n], labels)
        np.testing.assert_array_equal(r.path[i, :n], path)


@pytest.mark.parametrize("kernel", KERNELS)
def test_beam_traceback_segment_boundaries(fcd, kernel):
    """Many final depths around multiples of 64 (the wave kernels walk the labelling in 64-node
    segments along jump pointers; beam entries whose depths straddle a segment boundary at the
    end of the read were a bug once)."""
    x = gen_batch(15, 200, 340, 5)
    lengths = (140 + np.arange(200)).astype(np.int64)
    check_beam(fcd, x, 5, 0.1, lengths=lengths, kernel=kernel)


@pytest.mark.parametrize("kernel", KERNELS)
def test_beam_full_size_reads(fcd, kernel):
    """BASELINE config 2 shape (T=4000, N=5, beam 5, thr 0.1) on a handful of reads."""
    x = gen_batch(1, 8, 4000, 5)
    check_beam(fcd, x, 5, 0.1, kernel=kernel)


def test_beam_many_reads_chunked(fcd):
    """More reads than one workspace chunk holds: the C ABI decodes in chunks."""
    from fast_ctc_decode_amd import _native as nat
    x = gen_batch(14, 70, 300, 5)
    h = nat.default_handle()
    h.set_workspace_limit(8 << 20)
    try:
        check_beam(fcd, x, 5, 0.1, kernel=fcd.KERNEL_WAVE)
        check_beam(fcd, x, 5, 0.1, kernel=fcd.KERNEL_WAVE1)
        check_beam(fcd, x, 5, 0.1, kernel=fcd.KERNEL_GENERIC)
    finally:
        h.set_workspace_limit(0)


def test_viterbi_random(fcd):
    for N, collapse in ((5, True), (5, False), (3, True), (12, True)):
        x = gen_batch(20 + N, 7, 1000, N)
        x[0, 100:400, 1:] = 0.0  # a long blank stretch
        x[1, 10:300, 2] = 5.0    # a long single-label run spanning tiles
        lengths = np.array([1000, 1000, 1, 63, 64, 65, 999], np.int64)
        r = fcd.viterbi_search_batch_raw(x, collapse, lengths=lengths, qual=True)
        for i in range(x.shape[0]):
            xi = np.ascontiguousarray(x[i, :lengths[i]])
            labels, path, quals = oracle.viterbi_search_raw(xi, collapse)
            n = int(r.out_len[i])
            assert n == len(labels)
            np.testing.assert_array_equal(r.labels[i, :n], labels)
            np.testing.assert_array_equal(r.path[i, :n], path)
            got = [oracle.lib.fcdo_phred(float(q), 1.0, 0.0) for q in r.qual[i, :n]]
            assert [ord(c) for c in got] == list(quals)


def test_viterbi_qual_bits(fcd):
    """The per-run mean must be the reference's sequential f32 sum, bit for bit."""
    x = gen_batch(31, 4, 777, 5)
    r = fcd.viterbi_search_batch_raw(x, True, qual=True)
    for i in range(4):
        prob = x[i].max(1)
        lab = x[i].argmax(1)
        exp = []
        tot, cnt, last = np.float32(0), 0, -1
        for t in range(x.shape[1]):
            if lab[t] != 0 and last != lab[t]:
                if cnt:
                    exp.append(tot / np.float32(cnt))
                    tot, cnt = np.float32(0), 0
            if lab[t] != 0:
                tot = np.float32(tot + prob[t])
                cnt += 1
            last = lab[t]
        if cnt:
            exp.append(tot / np.float32(cnt))
        n = int(r.out_len[i])
        np.testing.assert_array_equal(r.qual[i, :n].view(np.uint32),
                                      np.array(exp, np.float32).view(np.uint32))


def to_dev(x):
    """A device tensor when a GPU is present (zero-copy *_dev entry points), the numpy array otherwise
    (host-staged *_host entry points: what FCD_TEST_EMU / tests/test_emu_parity.py run)."""
    try:
        import torch
        if torch.cuda.is_available():
            return torch.from_numpy(np.ascontiguousarray(x)).cuda()
    except ImportError:
        pass
    return x


def gen_crf(seed, B, T, S=4, N=5):
    rng = np.random.default_rng(seed)
    x = rng.random((B, T, S, N), dtype=np.float32)
    x /= x.sum(-1, keepdims=True)
    init = np.zeros((B, S), np.float32)
    init[np.arange(B), rng.integers(0, S, B)] = 1.0
    return x.astype(np.float32), init


@pytest.mark.parametrize("beam,thr", [(5, 0.0), (5, 0.1), (16, 0.05)])
def test_crf_beam_random(fcd, beam, thr):
    x, init = gen_crf(40 + beam, 5, 300)
    got = fcd.crf_beam_search_batch(x, init, "NACGT", beam, thr)
    for i in range(x.shape[0]):
        want = oracle.crf_beam_search(x[i], init[i], "NACGT", beam, thr)
        assert got[i] == want


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("beam,thr", [(1, 0.0), (5, 0.0), (5, 0.1), (8, 0.05), (12, 0.0), (10, 0.1)])
def test_crf_beam_kernels(fcd, kernel, beam, thr):
    """crf_beam_search on every kernel family (S = 4 states x 5 symbols is the wave kernels' shape)."""
    x, init = gen_crf(70 + beam, 9, 500)
    init[3] = [0.1, 0.7, 0.7, 0.05]      # a tie: the first maximum is the start state
    lengths = np.array([500, 499, 1, 0, 64, 65, 500, 31, 500], np.int64)
    r = fcd.crf_beam_search_batch_raw(to_dev(x), init, beam, thr, lengths=lengths, kernel=kernel).cpu()
    for i in range(x.shape[0]):
        n = int(r.out_len[i])
        if lengths[i] == 0:
            assert n == 0 and int(r.status[i]) == 0
            continue
        try:
            want = oracle.crf_beam_search(np.ascontiguousarray(x[i, :lengths[i]]), init[i], "NACGT", beam, thr)
        except RuntimeError as e:
            assert fcd.api.nat.status_string(int(r.status[i])) == str(e)
            continue
        assert int(r.status[i]) == 0
        seq = "".join("NACGT"[l] for l in r.labels[i, :n])
        assert (seq, r.path[i, :n].tolist()) == want


@pytest.mark.parametrize("S", [8, 16, 64, 256, 1024])
@pytest.mark.parametrize("beam,thr", [(1, 0.0), (5, 0.0), (5, 0.1), (12, 0.05), (16, 0.05), (32, 0.1), (64, 0.0)])
def test_crf_beam_many_states(fcd, S, beam, thr):
    """crf_beam_search with S = 4^k-style state counts (real basecaller heads), 5 symbols.  The
    register kernels gather only the rows of visited states (wave: beam <= 12, lane: beam <= 64); every
    kernel that accepts the shape must reproduce the oracle bit for bit, ragged lengths included, and the
    tie instrument must agree."""
    B, T = 6, 260
    x, init = gen_crf(7000 + S + beam, B, T, S=S)
    init[2] = 0.25                        # all tied: the first state starts
    lengths = np.array([T, T - 1, 1, 0, 64, 65], np.int64)
    want = []
    for i in range(B):
        L = int(lengths[i])
        want.append(None if L == 0 else
                    oracle.crf_beam_search_ambiguous(np.ascontiguousarray(x[i, :L]), init[i], beam, thr))
    kernels = [0, 1] + ([2, 3] if beam <= 12 else []) + [4]
    xd = to_dev(x)
    for kernel in kernels:
        r = fcd.crf_beam_search_batch_raw(xd, init, beam, thr, lengths=lengths, kernel=kernel,
                                          count_ambiguous=True).cpu()
        for i in range(B):
            n = int(r.out_len[i])
            if want[i] is None:
                assert n == 0 and int(r.status[i]) == 0
                continue
            st, labels, path, n_amb = want[i]
            assert int(r.status[i]) == st, (kernel, i)
            if st == 0:
                np.testing.assert_array_equal(r.labels[i, :n], labels)
                np.testing.assert_array_equal(r.path[i, :n], path)
            assert tuple(int(v) for v in r.ambiguous[i]) == n_amb, (kernel, i)


def test_crf_bad_state_parity(fcd):
    """A state count that lets (state * n_base) % n_state + label leave the table (the reference aborts on
    the ndarray bounds check, src/search.rs:72): only the LDS kernel accepts such shapes and reports
    BAD_STATE where the oracle reports the panic; init_state entries beyond S start out of range."""
    rng = np.random.default_rng(71)
    for S, n_init in ((5, 5), (6, 6), (16, 20), (64, 64)):
        B, T = 4, 60
        x = rng.random((B, T, S, 5), dtype=np.float32)
        x /= x.sum(-1, keepdims=True)
        init = rng.random((B, n_init)).astype(np.float32)
        if n_init > S:
            init[1, S + 1] = 2.0          # argmax beyond the table: out of range before the first row
        r = fcd.crf_beam_search_batch_raw(to_dev(x), init, 5, 0.0).cpu()
        for i in range(B):
            try:
                want = oracle.crf_beam_search(x[i], init[i], "NACGT", 5, 0.0)
            except RuntimeError as e:
                assert "panic" in str(e) and int(r.status[i]) == fcd.api.nat.ST_BAD_STATE, (S, i)
                continue
            n = int(r.out_len[i])
            assert int(r.status[i]) == 0
            assert ("".join("NACGT"[l] for l in r.labels[i, :n]), r.path[i, :n].tolist()) == want


def test_crf_greedy_random(fcd):
    x, init = gen_crf(50, 3, 400)
    for i in range(3):
        for q in (False, True):
            assert fcd.crf_greedy_search(x[i], init[i], "NACGT", q) == \
                oracle.crf_greedy_search(x[i], init[i], "NACGT", q)


def test_torch_device_path(fcd):
    """Zero-copy *_dev entry points on torch tensors: same answer as the host-staged path."""
    torch = pytest.importorskip("torch")
    x = gen_batch(60, 16, 500, 5)
    xd = torch.from_numpy(x).cuda()
    r = fcd.beam_search_batch_raw(xd, 5, 0.1)
    torch.cuda.synchronize()
    rh = fcd.beam_search_batch_raw(x, 5, 0.1)
    rc = r.cpu()
    np.testing.assert_array_equal(rc.out_len, rh.out_len)
    for i in range(16):
        n = int(rh.out_len[i])
        np.testing.assert_array_equal(rc.labels[i, :n], rh.labels[i, :n])
        np.testing.assert_array_equal(rc.path[i, :n].astype(np.uint32), rh.path[i, :n])
    v = fcd.viterbi_search_batch_raw(xd).cpu()
    vh = fcd.viterbi_search_batch_raw(x)
    np.testing.assert_array_equal(v.out_len, vh.out_len)


def test_concurrent_python_threads(fcd):
    """The reference releases the GIL and is re-entrant (src/lib.rs:199,353): many Python threads
    may decode at once.  Each thread gets its own fcd_handle (stream + workspace)."""
    import threading
    import fast_ctc_decode as compiled
    x = gen_batch(80, 24, 600, 5)
    want = [oracle.beam_search(x[i], "NACGT", 5, 0.1) for i in range(24)]
    got = [None] * 24
    errors = []

    def work(tid):
        try:
            for i in range(tid, 24, 6):
                m = compiled if (i % 2) else fcd  # both host layers, interleaved
                got[i] = m.beam_search(x[i], "NACGT", 5, 0.1)
                assert m.viterbi_search(x[i], "NACGT") == oracle.viterbi_search(x[i], "NACGT")
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    assert got == want


def test_c_abi_misuse_is_reported(fcd):
    """API misuse returns negative FCD_E_* codes with a message, never crashes."""
    import ctypes as C
    from fast_ctc_decode_amd import _native as nat
    h = nat.default_handle()
    x = gen_batch(81, 2, 50, 5)
    out = fcd.api._HostOut(2, 50)
    b = fcd.api._host_batch(x, False)
    lib = h.lib
    assert lib.fcd_beam_search_host(h.ptr, C.byref(b), 0, 0.1, 1, 0, C.byref(out.res)) == nat.E_INVALID
    assert b"beam_size" in lib.fcd_last_error(h.ptr)
    assert lib.fcd_beam_search_host(h.ptr, None, 5, 0.1, 1, 0, C.byref(out.res)) == nat.E_INVALID
    small = fcd.api._HostOut(2, 10)  # out_stride < T
    assert lib.fcd_beam_search_host(h.ptr, C.byref(b), 5, 0.1, 1, 0, C.byref(small.res)) == nat.E_INVALID
    # a shape the register kernel does not implement must be refused when it is forced
    x12 = gen_batch(82, 2, 50, 12)
    b12 = fcd.api._host_batch(x12, False)
    assert lib.fcd_beam_search_host(h.ptr, C.byref(b12), 5, 0.0, 1, nat.KERNEL_WAVE, C.byref(out.res)) == nat.E_UNSUPPORTED
    assert lib.fcd_beam_search_host(h.ptr, C.byref(b12), 5, 0.0, 1, nat.KERNEL_AUTO, C.byref(out.res)) == nat.OK
    # huge beam sizes fall outside the LDS budget: refused, not mis-computed
    assert lib.fcd_beam_search_host(h.ptr, C.byref(b), 100000, 0.0, 1, 0, C.byref(out.res)) == nat.E_UNSUPPORTED
    # and the handle is still usable afterwards
    assert lib.fcd_beam_search_host(h.ptr, C.byref(b), 5, 0.1, 1, 0, C.byref(out.res)) == nat.OK
    st, labels, path, _ = oracle.beam_search_raw(x[0], 5, 0.1)
    n = int(out.out_len[0])
    np.testing.assert_array_equal(out.labels[0, :n], labels)


def test_ragged_list_batch_equals_single_calls(fcd):
    """A Python list of reads of different lengths decodes like one call per read."""
    rng = np.random.default_rng(90)
    reads = [reference_style_rows(rng, int(t), 5) for t in (120, 1, 333, 64, 200)]
    got = fcd.beam_search_batch(reads, "NACGT", 5, 0.1)
    assert got == [oracle.beam_search(x, "NACGT", 5, 0.1) for x in reads]
    gv = fcd.viterbi_search_batch(reads, "NACGT", qstring=True)
    assert gv == [oracle.viterbi_search(x, "NACGT", True) for x in reads]


def test_dlpack_device_input(fcd):
    """Any device array speaking DLPack is accepted zero-copy (here: a torch tensor behind a
    minimal DLPack-only wrapper)."""
    torch = pytest.importorskip("torch")

    class DL:
        def __init__(self, t):
            self.t = t

        def __dlpack__(self, stream=None):
            return self.t.__dlpack__()

        def __dlpack_device__(self):
            return self.t.__dlpack_device__()

    x = gen_batch(91, 5, 300, 5)
    r = fcd.beam_search_batch_raw(DL(torch.from_numpy(x).cuda()), 5, 0.1).cpu()
    for i in range(5):
        st, labels, path, _ = oracle.beam_search_raw(x[i], 5, 0.1)
        n = int(r.out_len[i])
        np.testing.assert_array_equal(r.labels[i, :n], labels)
        np.testing.assert_array_equal(r.path[i, :n], path)


def _fuzz_case(seed):
    """One random (shape, parameters, input style) draw; quantised styles create exact f32 ties."""
    rng = np.random.default_rng(seed)
    N = int(rng.integers(2, 9))
    beam = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 11, 16]))
    T = int(rng.integers(1, 180))
    B = int(rng.integers(1, 6))
    thr = float(rng.choice([0.0, 0.0, 0.01, 0.1, 0.2]))
    collapse = bool(rng.integers(0, 2))
    style = int(rng.integers(0, 4))
    if style == 0:
        x = reference_style_rows(rng, B * T, N).reshape(B, T, N)
    elif style == 1:  # peaky softmax
        z = rng.normal(size=(B, T, N)).astype(np.float32) * 3.0
        e = np.exp(z - z.max(-1, keepdims=True))
        x = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    elif style == 2:  # coarse grid: many exactly equal probabilities (ties in the prune)
        x = (rng.integers(0, 4, size=(B, T, N)) / 4.0).astype(np.float32)
    else:  # powers of two incl. zeros: exact products, ties and dead candidates
        x = np.ldexp(1.0, -rng.integers(0, 5, size=(B, T, N))).astype(np.float32)
        x[rng.random((B, T, N)) < 0.15] = 0.0
    lengths = None
    if rng.integers(0, 3) == 0:
        lengths = rng.integers(0, T + 1, size=B).astype(np.int64)
    return x, beam, thr, collapse, lengths


@pytest.mark.parametrize("chunk", range(8))
def test_beam_fuzz(fcd, chunk):
    """Randomised differential test: every kernel family that supports the drawn shape must agree
    with the oracle bit for bit (labels, path, status), including inputs built to tie."""
    for seed in range(1000 + chunk * 12, 1000 + (chunk + 1) * 12):
        x, beam, thr, collapse, lengths = _fuzz_case(seed)
        for kernel in (0, 1, 2, 3, 4):
            try:
                check_beam(fcd, x, beam, thr, collapse, lengths=lengths, kernel=kernel)
            except RuntimeError as e:
                # an explicitly requested register kernel refuses shapes it is not built for
                assert kernel in (2, 3, 4) and " kernel: " in str(e), (seed, kernel, str(e))
            except AssertionError as e:
                raise AssertionError("fuzz seed %d kernel %d beam %d thr %g collapse %s shape %s: %s"
                                     % (seed, kernel, beam, thr, collapse, x.shape, e))


def crf_fuzz_seed(fcd, seed):
    """One random (S, N, beam, thr, T, init, ragged) draw of crf_beam_search on every kernel family."""
    if True:
        rng = np.random.default_rng(seed)
        shape = int(rng.integers(0, 3))
        if shape == 0:
            S, N = 4, 5                                            # the wave kernels' register-FIFO shape
        elif shape == 1:
            S, N = int(rng.choice([8, 16, 64, 256, 1024])), 5     # gathered rows (wave / lane kernels)
        else:
            S, N = int(rng.integers(1, 7)), int(rng.integers(2, 7))
        beam = int(rng.choice([1, 2, 5, 8, 12, 20, 40]))
        thr = float(rng.choice([0.0, 0.0, 0.05, 0.2]))
        B, T = int(rng.integers(1, 5)), int(rng.integers(1, 150))
        style = int(rng.integers(0, 3))
        if style == 0:
            x = rng.random((B, T, S, N), dtype=np.float32)
            x /= x.sum(-1, keepdims=True)
        elif style == 1:
            x = (rng.integers(0, 4, size=(B, T, S, N)) / 4.0).astype(np.float32)  # exact ties
        else:
            x = np.ldexp(1.0, -rng.integers(0, 6, size=(B, T, S, N))).astype(np.float32)
            x[rng.random((B, T, S, N)) < 0.1] = 0.0
        init = rng.random((B, S)).astype(np.float32)
        if rng.integers(0, 2):
            init = np.round(init * 2) / 2  # tied maxima: the first one is the start state
        init = init.astype(np.float32)
        lengths = rng.integers(0, T + 1, size=B).astype(np.int64) if rng.integers(0, 3) == 0 else None
        alpha = "N" + "ACGTUV"[:N - 1]
        for kernel in (0, 1, 2, 3, 4):
            try:
                r = fcd.crf_beam_search_batch_raw(to_dev(x), init, beam, thr, lengths=lengths, kernel=kernel).cpu()
            except RuntimeError as e:
                assert kernel in (2, 3, 4) and " kernel: " in str(e), (seed, kernel, str(e))
                continue
            for i in range(B):
                Ti = T if lengths is None else int(lengths[i])
                n = int(r.out_len[i])
                ctx = (seed, kernel, i, S, N, beam, thr, Ti)
                if Ti == 0:
                    assert n == 0 and int(r.status[i]) == 0, ctx
                    continue
                try:
                    want = oracle.crf_beam_search(np.ascontiguousarray(x[i, :Ti]), init[i], alpha, beam, thr)
                except RuntimeError as e:
                    if "panic" in str(e):
                        # an out-of-range transition state: the reference aborts the process
                        # (src/search.rs:72 ndarray bounds check); the library reports BAD_STATE
                        assert int(r.status[i]) == fcd.api.nat.ST_BAD_STATE, ctx
                    else:
                        assert fcd.api.nat.status_string(int(r.status[i])) == str(e), ctx
                    continue
                assert int(r.status[i]) == 0, ctx
                seq = "".join(alpha[l] for l in r.labels[i, :n])
                assert (seq, r.path[i, :n].tolist()) == want, ctx


@pytest.mark.parametrize("chunk", range(4))
def test_crf_fuzz(fcd, chunk):
    for seed in range(3000 + chunk * 10, 3000 + (chunk + 1) * 10):
        crf_fuzz_seed(fcd, seed)


def viterbi_fuzz_seed(fcd, seed):
    """Random shapes with quantised rows: argmax ties (first maximum wins), run means, ragged lengths."""
    if True:
        rng = np.random.default_rng(seed)
        N = int(rng.integers(2, 10))
        B, T = int(rng.integers(1, 6)), int(rng.integers(1, 700))
        collapse = bool(rng.integers(0, 2))
        if rng.integers(0, 2):
            x = (rng.integers(0, 3, size=(B, T, N)) / 2.0).astype(np.float32)
        else:
            x = reference_style_rows(rng, B * T, N).reshape(B, T, N)
        lengths = rng.integers(0, T + 1, size=B).astype(np.int64) if rng.integers(0, 2) else None
        r = fcd.viterbi_search_batch_raw(x, collapse, lengths=lengths, qual=True)
        for i in range(B):
            Ti = T if lengths is None else int(lengths[i])
            n = int(r.out_len[i])
            if Ti == 0:  # the reference asserts a non-empty matrix (:329); in a ragged batch: empty result
                assert n == 0, (seed, i)
                continue
            labels, path, quals = oracle.viterbi_search_raw(np.ascontiguousarray(x[i, :Ti]), collapse)
            assert n == len(labels), (seed, i)
            np.testing.assert_array_equal(r.labels[i, :n], labels)
            np.testing.assert_array_equal(r.path[i, :n], path)
            got = [oracle.lib.fcdo_phred(float(q), 1.0, 0.0) for q in r.qual[i, :n]]
            assert [ord(c) for c in got] == list(quals), (seed, i)


def test_viterbi_fuzz(fcd):
    for seed in range(4000, 4030):
        viterbi_fuzz_seed(fcd, seed)


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_viterbi_time_major_storage(fcd, dtype):
    """(T, B, N) storage -- what a basecaller network emits -- handed over as a (B, T, N) batch by its strides: the
    time-major kernel (8 reads per workgroup; the last B mod 8 reads on the strided kernel), ragged lengths, with and
    without quality values, vs the oracle on every read."""
    rng = np.random.default_rng(91)
    for B, T, N in ((37, 200, 5), (16, 64, 5), (48, 131, 3), (20, 65, 8)):
        xt = (rng.integers(0, 5, size=(T, B, N)) / 4.0).astype(np.float32)     # quantised: argmax ties
        xt[:, 1] = reference_style_rows(rng, T, N)
        xt[T // 2, 2, :] = np.nan
        xt = xt.astype(dtype)
        view = xt.transpose(1, 0, 2)                                            # (B, T, N), no copy
        assert not view.flags["C_CONTIGUOUS"]
        up = np.ascontiguousarray(view).astype(np.float32)
        lengths = rng.integers(0, T + 1, size=B).astype(np.int64)
        lengths[:3] = (T, T, T - 1)
        for collapse in (True, False):
            for lens, qual in ((None, False), (lengths, False), (lengths, True)):
                r = fcd.viterbi_search_batch_raw(view, collapse, lengths=lens, qual=qual)
                for i in range(B):
                    Ti = T if lens is None else int(lens[i])
                    n = int(r.out_len[i])
                    if Ti == 0:
                        assert n == 0
                        continue
                    labels, path, quals = oracle.viterbi_search_raw(np.ascontiguousarray(up[i, :Ti]), collapse)
                    assert n == len(labels), (B, T, N, collapse, i)
                    np.testing.assert_array_equal(r.labels[i, :n], labels)
                    np.testing.assert_array_equal(r.path[i, :n], path)
                    if qual:
                        got = [oracle.lib.fcdo_phred(float(q), 1.0, 0.0) for q in r.qual[i, :n]]
                        assert [ord(c) for c in got] == list(quals)


def test_viterbi_whole_tiles_without_quality(fcd):
    """The streaming kernel's whole-tile path (256-row tiles, no quality values: labels through a DPP wave shift):
    lengths on both sides of every tile and sub-tile boundary, runs and blank stretches crossing them, argmax ties,
    NaN rows, both element widths."""
    rng = np.random.default_rng(77)
    lens = [255, 256, 257, 319, 320, 321, 511, 512, 513, 767, 768, 1024, 1100, 1]
    B, T, N = len(lens), 1100, 5
    x = (rng.integers(0, 4, size=(B, T, N)) / 4.0).astype(np.float32)      # quantised: ties everywhere
    x[1] = reference_style_rows(rng, T, N)
    x[2, 200:600, 1:] = 0.0                                                # a blank stretch across two tiles
    x[3, 250:530, :] = 0.0
    x[3, 250:530, 3] = 1.0                                                 # one label for 280 rows
    x[4, 255, :] = np.nan                                                  # NaN rows at the seams
    x[4, 256, 2] = np.nan
    x[5, 63:66, 0] = np.nan
    lengths = np.array(lens, np.int64)
    for collapse in (True, False):
        for dtype in (np.float32, np.float16):
            xs = x.astype(dtype)
            up = xs.astype(np.float32)
            r = fcd.viterbi_search_batch_raw(xs, collapse, lengths=lengths)
            for i in range(B):
                labels, path, _ = oracle.viterbi_search_raw(np.ascontiguousarray(up[i, :lens[i]]), collapse)
                n = int(r.out_len[i])
                assert n == len(labels), (collapse, dtype, i)
                np.testing.assert_array_equal(r.labels[i, :n], labels)
                np.testing.assert_array_equal(r.path[i, :n], path)


def test_batch_sequences_path_flavours(fcd):
    """BatchResult.sequences: the vectorised string build and the three path flavours agree with the
    single-read calls; multi-character alphabets take the generic route."""
    x = gen_batch(55, 5, 120, 5)
    want = [fcd.beam_search(x[i], "NACGT", 5, 0.1) for i in range(5)]
    assert fcd.beam_search_batch(x, "NACGT", 5, 0.1) == want
    arr = fcd.beam_search_batch(x, "NACGT", 5, 0.1, paths="array")
    assert [(s, p.tolist()) for s, p in arr] == want
    none = fcd.beam_search_batch(x, "NACGT", 5, 0.1, paths=None)
    assert [s for s, _ in none] == [s for s, _ in want] and all(p is None for _, p in none)
    multi = ["N", "Ade", "Cyt", "Gua", "Thy"]
    wm = [fcd.beam_search(x[i], multi, 5, 0.1) for i in range(5)]
    assert fcd.beam_search_batch(x, multi, 5, 0.1) == wm


def test_half_precision_device_tensors(fcd):
    """float16 / bfloat16 device posteriors are upcast exactly: same result as the f32 call on the
    upcast matrix (which is what the reference would be given)."""
    torch = pytest.importorskip("torch")
    x = torch.from_numpy(gen_batch(66, 4, 300, 5)).cuda()
    for dt in (torch.float16, torch.bfloat16):
        xh = x.to(dt)
        up = xh.float()
        a = fcd.beam_search_batch_raw(xh, 5, 0.1).cpu()
        b = fcd.beam_search_batch_raw(up, 5, 0.1).cpu()
        np.testing.assert_array_equal(a.out_len, b.out_len)
        for i in range(4):
            n = int(a.out_len[i])
            np.testing.assert_array_equal(a.labels[i, :n], b.labels[i, :n])
            np.testing.assert_array_equal(a.path[i, :n], b.path[i, :n])
        va, vb = fcd.viterbi_search_batch_raw(xh).cpu(), fcd.viterbi_search_batch_raw(up).cpu()
        np.testing.assert_array_equal(va.out_len, vb.out_len)
        check_beam(fcd, up.cpu().numpy(), 5, 0.1)


def test_crf_greedy_batch(fcd):
    """The batched crf_greedy_search (host arrays and device tensors) equals the single-read calls."""
    torch = pytest.importorskip("torch")
    x, init = gen_crf(51, 5, 300)
    want = [fcd.crf_greedy_search(x[i], init[i], "NACGT") for i in range(5)]
    assert want == [oracle.crf_greedy_search(x[i], init[i], "NACGT", False) for i in range(5)]
    assert fcd.crf_greedy_search_batch(x, init, "NACGT") == want
    assert fcd.crf_greedy_search_batch(torch.from_numpy(x).cuda(), init, "NACGT") == want
    wq = [fcd.crf_greedy_search(x[i], init[i], "NACGT", True) for i in range(5)]
    assert fcd.crf_greedy_search_batch(x, init, "NACGT", qstring=True) == wq


def crf_greedy_fuzz_seed(fcd, seed):
    """One random draw of crf_greedy_search: small state counts take the streaming kernel (function
    composition scan), larger ones the serial walk; ties, NaNs, out-of-range transitions, ragged lengths."""
    rng = np.random.default_rng(seed)
    S = int(rng.choice([1, 2, 3, 4, 4, 5, 8, 9, 12]))
    N = int(rng.integers(2, 8))
    if S * N > 32 and rng.integers(0, 2):
        N = max(2, 32 // S)
    B, T = int(rng.integers(1, 5)), int(rng.integers(1, 400))
    style = int(rng.integers(0, 3))
    if style == 0:
        x = rng.random((B, T, S, N), dtype=np.float32)
    elif style == 1:
        x = (rng.integers(0, 3, size=(B, T, S, N)) / 2.0).astype(np.float32)   # ties: first maximum wins
    else:
        x = rng.random((B, T, S, N), dtype=np.float32)
        x[..., 0] += 0.7                                                       # mostly blanks
    if rng.integers(0, 6) == 0:
        x[rng.integers(0, B), rng.integers(0, T), rng.integers(0, S), rng.integers(0, N)] = np.nan
    init = rng.random((B, int(rng.choice([S, S, S + 2])))).astype(np.float32)
    lengths = rng.integers(0, T + 1, size=B).astype(np.int64) if rng.integers(0, 3) == 0 else None
    alpha = "N" + "ACGTUVW"[:N - 1]
    r = fcd.crf_greedy_search_batch_raw(x, init, lengths, qual=True)
    for i in range(B):
        Ti = T if lengths is None else int(lengths[i])
        ctx = (seed, i, S, N, Ti)
        if Ti == 0:
            assert int(r.out_len[i]) == 0, ctx
            continue
        try:
            seq, path = oracle.crf_greedy_search(np.ascontiguousarray(x[i, :Ti]), init[i], alpha, False)
        except RuntimeError:
            assert int(r.status[i]) == fcd.api.nat.ST_BAD_STATE, ctx
            continue
        assert int(r.status[i]) == 0, ctx
        n = int(r.out_len[i])
        assert "".join(alpha[l] for l in r.labels[i, :n]) == seq and r.path[i, :n].tolist() == path, ctx
        qs = oracle.crf_greedy_search(np.ascontiguousarray(x[i, :Ti]), init[i], alpha, True)[0][n:]
        assert "".join(oracle.phred(float(q)) for q in r.qual[i, :n]) == qs, ctx


def test_crf_greedy_time_major_storage(fcd):
    """(T, B, S, N) storage handed over as a (B, T, S, N) batch by its strides: the time-major instantiation of the
    streaming kernel (four neighbouring reads per workgroup; the last B mod 4 reads on the serial kernel), ragged lengths,
    quality values, a NaN and an out-of-range state -- vs the oracle on every read."""
    rng = np.random.default_rng(93)
    for B, T, S, N in ((21, 150, 4, 5), (8, 64, 4, 5), (12, 70, 2, 3), (9, 33, 8, 4)):
        xt = rng.random((T, B, S, N), dtype=np.float32)
        xt[:, 1] = (rng.integers(0, 4, size=(T, S, N)) / 4.0).astype(np.float32)   # argmax ties
        xt[T // 2, 2, :, :] = np.nan
        view = xt.transpose(1, 0, 2, 3)
        assert not view.flags["C_CONTIGUOUS"]
        init = rng.random((B, S), dtype=np.float32)
        lengths = rng.integers(0, T + 1, size=B).astype(np.int64)
        lengths[:3] = (T, T, T)
        for lens in (None, lengths):
            r = fcd.crf_greedy_search_batch_raw(view, init, lengths=lens, qual=True)
            for i in range(B):
                Ti = T if lens is None else int(lens[i])
                xi = np.ascontiguousarray(view[i, :Ti])
                if Ti == 0:
                    continue
                try:
                    seq, path = oracle.crf_greedy_search(xi, init[i], "NACGTUV"[:N], qstring=False)
                except RuntimeError:
                    assert int(r.status[i]) != 0, (B, T, S, N, i)
                    continue
                assert int(r.status[i]) == 0, (B, T, S, N, i)
                n = int(r.out_len[i])
                assert "".join("NACGTUV"[l] for l in r.labels[i, :n]) == seq
                np.testing.assert_array_equal(r.path[i, :n], path)


def test_crf_greedy_fuzz(fcd):
    for seed in range(7000, 7060):
        crf_greedy_fuzz_seed(fcd, seed)


def test_release_workspace(fcd):
    """fcd_release_workspace gives the arena / staging memory back; the next call allocates afresh and returns
    the same results."""
    from fast_ctc_decode_amd import _native as nat
    x = gen_batch(4242, 6, 300, 5)
    h = nat.default_handle()
    for beam in (5, 32):
        a = fcd.beam_search_batch_raw(x, beam, 0.1, True)
        h.release_workspace()
        b = fcd.beam_search_batch_raw(x, beam, 0.1, True)
        np.testing.assert_array_equal(a.out_len, b.out_len)
        np.testing.assert_array_equal(a.status, b.status)
        for i in range(len(x)):
            n = int(a.out_len[i])
            np.testing.assert_array_equal(a.labels[i, :n], b.labels[i, :n])
            np.testing.assert_array_equal(a.path[i, :n], b.path[i, :n])
    check_beam(fcd, x, 5, 0.1)
