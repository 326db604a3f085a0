"""duplex::beam_search (src/duplex.rs:443-650) on the GPU vs the CPU oracle.

The kernels define ln / exp / ln_1p as correctly rounded f32, so the exact
target is the oracle with FCDO_MATH_CR: strings must be IDENTICAL.  Against the oracle on the host
libm (what the reference computes on this machine's glibc) strings may differ in the rare case a
last-bit difference flips a near-tie; that agreement is checked with an explicit tolerance."""
import numpy as np
import pytest

import kat_cases
from kat_cases import reference_style_rows
from oracle import oracle

pytestmark = pytest.mark.gpu

LSE, MAX, CR = oracle.LOGSUMEXP, oracle.MAXMODE, oracle.MATH_CR


@pytest.fixture(scope="module")
def fcd():
    import fast_ctc_decode_amd as m
    return m


class _Mode:
    def __init__(self, fcd, mode):
        self.fcd, self.mode = fcd, mode

    def beam_search_duplex(self, *a, **k):
        return self.fcd.beam_search_duplex(*a, logadd_mode=self.mode, **k)


@pytest.mark.parametrize("mode", [0, 1], ids=["logsumexp", "max"])
@pytest.mark.parametrize("case", kat_cases.DUPLEX_CASES, ids=lambda f: f.__name__)
def test_kat_duplex(fcd, case, mode):
    case(_Mode(fcd, mode))


def band(T1, T2, w):
    i = np.arange(T1)
    return np.stack([np.maximum(0, i - w), np.minimum(T2, i + w)], 1).astype(np.uint64)


def pairs(seed, B, T1, T2, N=5):
    rng = np.random.default_rng(seed)
    x1 = reference_style_rows(rng, B * T1, N).reshape(B, T1, N)
    x2 = reference_style_rows(rng, B * T2, N).reshape(B, T2, N)
    return x1, x2


def oracle_strings(x1, x2, alpha, envs, beam, thr, collapse, mode):
    out = []
    for i in range(x1.shape[0]):
        e = None if envs is None else envs[i]
        try:
            out.append(oracle.beam_search_duplex(x1[i], x2[i], alpha, e, beam, thr, collapse, mode))
        except RuntimeError as err:
            out.append(str(err))
    return out


def gpu_strings(fcd, x1, x2, alpha, envs, beam, thr, collapse, mode):
    r = fcd.beam_search_duplex_batch_raw(x1, x2, envs, beam, thr, collapse, logadd_mode=mode).cpu()
    out = []
    for i in range(x1.shape[0]):
        if int(r.status[i]) != 0:
            out.append(fcd.api.nat.status_string(int(r.status[i])))
        else:
            out.append("".join(alpha[l] for l in r.labels[i, :int(r.out_len[i])]))
    return out


@pytest.mark.parametrize("mode", [LSE, MAX], ids=["logsumexp", "max"])
@pytest.mark.parametrize("collapse", [True, False])
def test_duplex_banded_exact(fcd, mode, collapse):
    x1, x2 = pairs(300 + mode, 6, 160, 150)
    envs = np.stack([band(160, 150, 24)] * 6)
    got = gpu_strings(fcd, x1, x2, "NACGT", envs, 5, 0.1, collapse, mode)
    want = oracle_strings(x1, x2, "NACGT", envs, 5, 0.1, collapse, mode | CR)
    assert got == want


@pytest.mark.parametrize("mode", [LSE, MAX], ids=["logsumexp", "max"])
def test_duplex_special_values_in_wide_windows(fcd, mode):
    """Posteriors of exactly 1 (log 0), exactly 0 (log -inf), above 1 and NaN inside a +-24 band: max mode's
    four-lanes-per-node loop runs its rows on v_max_f32 and must fall back to LogSpace::add's compare-and-select form
    for every group of rows such a value touches -- and for the groups AFTER it, whose incoming chain state may be a
    NaN or a zero.  Both modes, exact vs the oracle."""
    x1, x2 = pairs(340 + mode, 8, 120, 120)
    x2[0, 30:34, :] = 0.0
    x2[0, 30:34, 2] = 1.0            # probability one: its logarithm is 0
    x2[1, 50, 1] = 0.0               # probability zero: -inf
    x2[1, 51, :] = 0.0               # a whole row of zeros
    x2[2, 60:63, 3] = 1.5            # a "probability" above 1: a positive logarithm
    x2[3, 40, 0] = np.nan            # NaN in the blank column
    x2[4, 41, 2] = np.nan            # NaN in a label column
    x2[4, 0, 2] = np.nan             # ... and in the first row: every window that starts there is NaN from its first
                                     # row on for that label -- a state the following groups of rows must carry
    x1[5, 20, :] = 0.0
    x1[5, 20, 1] = 1.0               # read 1 too
    x2[6, 0, :] = 0.0
    x2[6, 0, 0] = 1.0                # at the very first row
    x2[7, 119, 4] = 1.0              # and at the very last
    envs = np.stack([band(120, 120, 24)] * 8)
    got = gpu_strings(fcd, x1, x2, "NACGT", envs, 5, 0.05, True, mode)
    want = oracle_strings(x1, x2, "NACGT", envs, 5, 0.05, True, mode | CR)
    assert got == want


@pytest.mark.parametrize("mode", [LSE, MAX], ids=["logsumexp", "max"])
def test_duplex_default_envelope_exact(fcd, mode):
    x1, x2 = pairs(310 + mode, 4, 60, 70)
    got = gpu_strings(fcd, x1, x2, "NACGT", None, 5, 0.0, True, mode)
    want = oracle_strings(x1, x2, "NACGT", None, 5, 0.0, True, mode | CR)
    assert got == want


@pytest.mark.parametrize("beam,N", [(1, 5), (3, 3), (8, 5), (16, 4)])
def test_duplex_shapes_exact(fcd, beam, N):
    alpha = "NACGTXY"[:N]
    x1, x2 = pairs(320 + beam, 4, 90, 100, N)
    envs = np.stack([band(90, 100, 16)] * 4)
    thr = 0.05
    got = gpu_strings(fcd, x1, x2, alpha, envs, beam, thr, True, LSE)
    want = oracle_strings(x1, x2, alpha, envs, beam, thr, True, LSE | CR)
    assert got == want


def test_duplex_envelope_errors_and_edges(fcd):
    x1, x2 = pairs(330, 5, 40, 40)
    envs = np.stack([band(40, 40, 8)] * 5)
    envs[1, 10, 0] = 30           # lower bound jumps past the previous upper bound -> InvalidEnvelope
    envs[2, 5] = (7, 7)           # empty row -> InvalidEnvelope
    envs[3, :, 1] = 1000          # upper bounds beyond read 2 are clamped ... but row 0 must be <= T2
    envs[3, 0, 1] = 8
    x1[4, 20:] = 0.0              # nothing passes the threshold -> RanOutOfBeam
    got = gpu_strings(fcd, x1, x2, "NACGT", envs, 5, 0.1, True, LSE)
    want = oracle_strings(x1, x2, "NACGT", envs, 5, 0.1, True, LSE | CR)
    assert got == want
    assert got[1] == got[2] == "Invalid envelope values"


def test_duplex_vs_host_libm_tolerance(fcd):
    """Tolerance statement: vs the oracle on the HOST libm (glibc logf/expf/log1pf, not correctly
    rounded) at least 90 % of random pairs must give the identical string."""
    x1, x2 = pairs(340, 20, 200, 200)
    envs = np.stack([band(200, 200, 32)] * 20)
    got = gpu_strings(fcd, x1, x2, "NACGT", envs, 5, 0.1, True, LSE)
    want = oracle_strings(x1, x2, "NACGT", envs, 5, 0.1, True, LSE)
    same = sum(g == w for g, w in zip(got, want))
    assert same >= 18, (same, 20)


def test_duplex_vs_host_libm_many_pairs(fcd):
    """The same tolerance statement on a sample large enough to put a number on it: 300 pairs (both searches,
    two envelope widths) against the oracle on the host's glibc.  The kernels define ln / exp / ln_1p as correctly
    rounded; glibc 2.35's are within 1 ULP, so a consensus differs only where a last-bit difference flips a
    near-tie -- observed on the MI355X: 300 of 300 identical."""
    same = total = 0
    for seed, n, T, w in ((350, 200, 120, 16), (351, 100, 160, 40)):
        x1, x2 = pairs(seed, n, T, T)
        envs = np.stack([band(T, T, w)] * n)
        got = gpu_strings(fcd, x1, x2, "NACGT", envs, 5, 0.1, True, LSE)
        want = oracle_strings(x1, x2, "NACGT", envs, 5, 0.1, True, LSE)
        same += sum(g == w_ for g, w_ in zip(got, want))
        total += n
    print("duplex vs host libm: %d of %d pairs identical" % (same, total))
    assert same >= 0.9 * total, (same, total)


def test_duplex_config5_shape_sample(fcd):
    """BASELINE config 5 shape (T1 = T2 = 2000, band +-64) on a few pairs, exact vs the CR oracle."""
    x1, x2 = pairs(4, 3, 2000, 2000)
    envs = np.stack([band(2000, 2000, 64)] * 3)
    got = gpu_strings(fcd, x1, x2, "NACGT", envs, 5, 0.1, True, LSE)
    want = oracle_strings(x1, x2, "NACGT", envs, 5, 0.1, True, LSE | CR)
    assert got == want


@pytest.mark.parametrize("mode", [LSE, MAX], ids=["logsumexp", "max"])
def test_duplex_config5_full_size(fcd, mode):
    """BASELINE config 5 at its stated size: 1024 pairs, T1 = T2 = 2000, band +-64, beam 5, threshold 0.1, both
    log-add modes.  Every pair decodes; reversing the batch reverses the answers (pairs are independent: the
    property that does not need an oracle); 64 pairs spread over the batch equal the correctly-rounded oracle (r06:
    eight before), eight of them with their tie counters; the slot-resident kernel (csrc/duplex_slots.hip, what AUTO
    runs here) and the any-shape kernel (csrc/duplex.hip) agree on every pair."""
    torch = pytest.importorskip("torch")
    n_pairs, T, n_oracle = 1024, 2000, 64
    x1, x2 = pairs(4, n_pairs, T, T)
    env = band(T, T, 64)
    x1d, x2d = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()
    envd = torch.from_numpy(np.broadcast_to(env, (n_pairs, T, 2)).copy().view(np.int64)).cuda()
    r = fcd.beam_search_duplex_batch_raw(x1d, x2d, envd, 5, 0.1, True, logadd_mode=mode, count_ambiguous=True).cpu()
    assert (np.asarray(r.status) == 0).all()
    rev = fcd.beam_search_duplex_batch_raw(x1d.flip(0).contiguous(), x2d.flip(0).contiguous(), envd, 5, 0.1, True,
                                           logadd_mode=mode).cpu()
    assert np.array_equal(np.asarray(rev.out_len)[::-1], np.asarray(r.out_len))
    lens = np.asarray(r.out_len).astype(np.int64)
    mask = np.arange(r.labels.shape[1])[None, :] < lens[:, None]
    assert np.array_equal(np.where(mask, r.labels, 0), np.where(mask, np.asarray(rev.labels)[::-1], 0))
    from concurrent.futures import ThreadPoolExecutor
    pick = np.linspace(0, n_pairs - 1, n_oracle).astype(np.int64)
    with ThreadPoolExecutor(32) as pool:  # (the oracle's C routine runs outside the interpreter lock)
        wants = list(pool.map(lambda i: oracle.beam_search_duplex(x1[i], x2[i], "NACGT", env, 5, 0.1, True, mode | CR), pick))
    for i, want in zip(pick, wants):
        assert "".join("NACGT"[l] for l in r.labels[i, :lens[i]]) == want, i
    for i in pick[::8]:  # (the counters are the last call's: one pair at a time)
        assert oracle.beam_search_duplex(x1[i], x2[i], "NACGT", env, 5, 0.1, True, mode | CR) is not None
        assert tuple(int(v) for v in r.ambiguous[i]) == oracle.duplex_last_ambiguous(), i
    # the two kernels, every pair
    from fast_ctc_decode_amd import _native as nat
    h = nat.default_handle()
    assert h.lib.fcd_debug_set_duplex_kernel(h.ptr, 1) == 0
    try:
        legacy = fcd.beam_search_duplex_batch_raw(x1d, x2d, envd, 5, 0.1, True, logadd_mode=mode, count_ambiguous=True).cpu()
    finally:
        assert h.lib.fcd_debug_set_duplex_kernel(h.ptr, 0) == 0
    assert np.array_equal(np.asarray(legacy.out_len), np.asarray(r.out_len)) and np.array_equal(np.asarray(legacy.status), np.asarray(r.status))
    assert np.array_equal(np.where(mask, legacy.labels, 0), np.where(mask, r.labels, 0))
    assert np.array_equal(np.asarray(legacy.ambiguous), np.asarray(r.ambiguous))
    amb = np.asarray(r.ambiguous)
    print("config 5 %s: pairs with a > 20-candidate kept tie %d, with a result-changing tie %d, both %d of %d"
          % ("max" if mode == MAX else "logsumexp", int((amb[:, 0] > 0).sum()), int((amb[:, 1] > 0).sum()),
             int(((amb[:, 0] > 0) & (amb[:, 1] > 0)).sum()), n_pairs))


@pytest.mark.parametrize("which", [1, 2], ids=["any-shape", "slot-resident"])
def test_duplex_each_kernel_forced(fcd, which):
    """The duplex searches have two kernels since r06 -- csrc/duplex_slots.hip (beam_size * N <= 64 and its rings in LDS:
    what AUTO picks for every shape of this file but the very wide ones) and csrc/duplex.hip (any shape).  Forced one at
    a time (fcd_debug_set_duplex_kernel), each passes the exact comparisons with the oracle: banded, special values,
    shapes, receding bounds, CRF, ties under both orders."""
    from fast_ctc_decode_amd import _native as nat
    h = nat.default_handle()
    assert h.lib.fcd_debug_set_duplex_kernel(h.ptr, which) == 0
    try:
        test_duplex_banded_exact(fcd, LSE, True)
        test_duplex_banded_exact(fcd, MAX, False)
        test_duplex_special_values_in_wide_windows(fcd, MAX)
        test_duplex_shapes_exact(fcd, 3, 3)
        test_duplex_shapes_exact(fcd, 8, 5)
        test_duplex_receding_upper_bound(fcd, LSE)
        test_duplex_envelope_errors_and_edges(fcd)
        test_duplex_tie_counters(fcd, MAX)
        for seed in (7000, 100369):
            assert special_values_case(fcd, seed, MAX) and special_values_case(fcd, seed, LSE)
        x1, i1, x2, i2 = crf_pairs(405, 70, 64)
        env = band(70, 64, 20)
        for mode in (LSE, MAX):
            want = oracle.crf_beam_search_duplex(x1, i1, x2, i2, "NACGT", env, 5, 0.1, mode | CR)
            assert fcd.crf_beam_search_duplex(x1, i1, x2, i2, "NACGT", env, 5, 0.1, logadd_mode=mode) == want
    finally:
        assert h.lib.fcd_debug_set_duplex_kernel(h.ptr, 0) == 0


def test_duplex_wobbly_envelope_exact(fcd):
    """Envelopes whose bounds do not slide monotonically (lower bound moving back, upper bound
    jumping by several rows, plateaus) exercise both the incremental and the rescanning paths of
    extend_secondary_probs / update_max."""
    rng = np.random.default_rng(350)
    T1 = T2 = 140
    x1, x2 = pairs(350, 6, T1, T2)
    envs = []
    for p in range(6):
        lo = np.zeros(T1, np.int64)
        hi = np.zeros(T1, np.int64)
        cur_lo, cur_hi = 0, 10 + int(rng.integers(0, 10))
        for t in range(T1):
            cur_hi = min(T2, cur_hi + int(rng.choice([0, 1, 1, 1, 2, 5])))
            cur_lo = max(0, min(cur_hi - 1, cur_lo + int(rng.choice([-3, 0, 1, 1, 1, 2]))))
            lo[t], hi[t] = cur_lo, cur_hi
        envs.append(np.stack([lo, hi], 1).astype(np.uint64))
    envs = np.stack(envs)
    for mode in (LSE, MAX):
        got = gpu_strings(fcd, x1, x2, "NACGT", envs, 5, 0.1, True, mode)
        want = oracle_strings(x1, x2, "NACGT", envs, 5, 0.1, True, mode | CR)
        assert got == want


@pytest.mark.parametrize("mode", [LSE, MAX], ids=["logsumexp", "max"])
def test_duplex_receding_upper_bound(fcd, mode):
    """The reference keeps the PREVIOUS row's upper bound (src/duplex.rs:524), not the largest seen: after the bound
    moved back, moving forward again by less than a beam entry's window already covers trips
    `assert!(current_end < upper_bound)` (:363-366) and aborts.  The oracle reports that as a panic (found by the
    naive cross-check, tests/test_naive_crosscheck.py), the kernels as FCD_ST_BAD_STATE; envelopes that recede
    without tripping the assertion decode to the oracle's strings."""
    n, T = 16, 60
    x1, x2 = pairs(77 + mode, n, T, T)
    rng = np.random.default_rng(5)
    envs = np.zeros((n, T, 2), np.uint64)
    for p in range(n):
        base = np.minimum(T, np.arange(T) + 7)
        hi = base.copy()
        for t in rng.choice(np.arange(3, T - 12), 3, replace=False):
            hi[t] = base[t - 1] - 1                      # the bound moves back by one row ...
            if p % 2:
                hi[t + 1] = base[t - 1]                  # ... and returns to where the windows already end: abort
        envs[p, :, 0] = np.maximum(0, np.arange(T) - 6)
        envs[p, :, 1] = hi
    r = fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.05, True, logadd_mode=mode).cpu()
    panics = 0
    for i in range(n):
        try:
            want = oracle.beam_search_duplex(x1[i], x2[i], "NACGT", envs[i], 5, 0.05, True, mode | CR)
            assert int(r.status[i]) == 0, i
            assert "".join("NACGT"[l] for l in r.labels[i, :int(r.out_len[i])]) == want, i
        except RuntimeError as e:
            assert "panic" in str(e), e
            assert int(r.status[i]) == fcd.api.nat.ST_BAD_STATE, i
            panics += 1
    assert 0 < panics < n  # both outcomes occur
    with pytest.raises(RuntimeError, match="the reference would abort"):
        bad = [i for i in range(n) if int(r.status[i]) != 0][0]
        fcd.beam_search_duplex(x1[bad], x2[bad], "NACGT", envs[bad], 5, 0.05, logadd_mode=mode)


def test_duplex_wide_band_unstaged_path(fcd):
    """A band too wide for the LDS tile (beam 16, +-300 rows) takes the HBM-only path."""
    x1, x2 = pairs(360, 2, 400, 400)
    envs = np.stack([band(400, 400, 300)] * 2)
    got = gpu_strings(fcd, x1, x2, "NACGT", envs, 16, 0.1, True, MAX)
    want = oracle_strings(x1, x2, "NACGT", envs, 16, 0.1, True, MAX | CR)
    assert got == want


def test_logspace_arithmetic_bits(fcd):
    """LogSpace::new / LogSpace::add of the duplex kernels vs the oracle's correctly rounded
    versions, bit for bit, on a million operand pairs incl. the edge cases (-inf, equal, NaN,
    far apart, denormal probabilities)."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    from fast_ctc_decode_amd import _native as nat
    rng = np.random.default_rng(370)
    n = 1 << 20
    pa = rng.random(n, dtype=np.float32) ** 8           # probabilities, many tiny
    pb = rng.random(n, dtype=np.float32) ** 3
    pa[:8] = [0.0, 1.0, 1e-45, 1e-38, 0.5, 0.25, 3.0, np.nan]
    a = np.log(pa.astype(np.float64)).astype(np.float32)
    b = np.log(pb.astype(np.float64)).astype(np.float32)
    b[8:4096] = a[8:4096] - rng.random(4088, dtype=np.float32) * 120.0   # wide range of small-big
    b[4096:4200] = a[4096:4200]                                           # equal operands
    a[4200:4210] = -np.inf
    b[4205:4215] = -np.inf
    ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    pad = torch.from_numpy(pa).cuda()
    out_add, out_ln = torch.empty_like(ad), torch.empty_like(ad)
    h = nat.default_handle()
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    # a dense sweep of small - big over [-110, 0] against assorted `big` (incl. 0.0, tiny, huge): every
    # branch of the fast path (exp -> 0, |big| shortcut, subnormal exp, ln_1p(e) = e, both Ziv fall-backs)
    m = 1 << 19
    a[8192:8192 + m] = rng.choice(np.array([0.0, -1e-30, -1e-3, -0.7, -5.0, -88.0, -1e4, 3.5, 1e30], np.float32), m)
    b[8192:8192 + m] = a[8192:8192 + m] - (rng.random(m, dtype=np.float32) ** 2) * np.float32(110.0)
    ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()

    def oracle_add(x, y, omode):
        x, y = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32)
        out = np.empty_like(x)
        oracle.lib.fcdo_logspace_add_batch(x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size, omode)
        return out

    # (+ 4: the same flavours in the form the window-building loop evaluates them; 2 / 6: on glibc 2.35's expf and
    # log1pf, bit for bit -- compared with the oracle on the HOST'S libm, which in the build image is that library)
    glibc235 = __import__("platform").libc_ver() == ("glibc", "2.35")
    for mode, omode in ((0, LSE | CR), (1, MAX | CR), (4, LSE | CR), (5, MAX | CR)) + (((2, LSE), (6, LSE)) if glibc235 else ()):
        h.check(h.lib.fcd_logspace_probe_dev(h.ptr, ad.data_ptr(), bd.data_ptr(), out_add.data_ptr(),
                                             out_ln.data_ptr(), n, mode))
        torch.cuda.synchronize()
        g = out_add.cpu().numpy()
        want = oracle_add(a, b, omode)      # every one of the 2^20 operand pairs
        same = (g.view(np.uint32) == want.view(np.uint32)) | (np.isnan(g) & np.isnan(want))
        bad = np.flatnonzero(~same)
        assert same.all(), (mode, bad.size, bad[:5], a[bad[:5]], b[bad[:5]], g[bad[:5]], want[bad[:5]])
    # ln of the posteriors: probe ln(pa) against the correctly rounded reference
    h.check(h.lib.fcd_logspace_probe_dev(h.ptr, pad.data_ptr(), bd.data_ptr(), out_add.data_ptr(),
                                         out_ln.data_ptr(), n, 0))
    torch.cuda.synchronize()
    got_ln = out_ln.cpu().numpy()
    with np.errstate(divide="ignore", invalid="ignore"):
        want_ln = np.log(pa.astype(np.longdouble)).astype(np.float32)
    same = (got_ln.view(np.uint32) == want_ln.view(np.uint32)) | (np.isnan(got_ln) & np.isnan(want_ln))
    assert same.all(), (int((~same).sum()), got_ln[~same][:5], want_ln[~same][:5])


def crf_pairs(seed, T1, T2, S=4, N=5):
    rng = np.random.default_rng(seed)
    x1 = rng.random((T1, S, N), dtype=np.float32)
    x2 = rng.random((T2, S, N), dtype=np.float32)
    x1 /= x1.sum(-1, keepdims=True)
    x2 /= x2.sum(-1, keepdims=True)
    i1 = np.zeros(S, np.float32)
    i2 = np.zeros(S, np.float32)
    i1[rng.integers(0, S)] = 1.0
    i2[rng.integers(0, S)] = 1.0
    return x1.astype(np.float32), i1, x2.astype(np.float32), i2


@pytest.mark.parametrize("mode", [LSE, MAX], ids=["logsumexp", "max"])
@pytest.mark.parametrize("beam,thr,w", [(5, 0.0, 12), (5, 0.1, 20), (3, 0.05, None), (8, 0.0, 16)])
def test_crf_duplex_exact(fcd, mode, beam, thr, w):
    """duplex::crf_beam_search (src/duplex.rs:652-834) vs the correctly rounded oracle, both host layers."""
    import fast_ctc_decode as compiled
    for seed in (400, 401, 402):
        x1, i1, x2, i2 = crf_pairs(seed + beam, 70, 64)
        env = None if w is None else band(70, 64, w)
        want = oracle.crf_beam_search_duplex(x1, i1, x2, i2, "NACGT", env, beam, thr, mode | CR)
        got = fcd.crf_beam_search_duplex(x1, i1, x2, i2, "NACGT", env, beam, thr, logadd_mode=mode)
        assert got == want
        compiled._set_duplex_logadd_mode("max" if mode == MAX else "logsumexp")
        try:
            assert compiled.crf_beam_search_duplex(x1, i1, x2, i2, "NACGT", env, beam, thr) == want
        finally:
            compiled._set_duplex_logadd_mode("logsumexp")


def test_crf_duplex_multichar_and_errors(fcd):
    x1, i1, x2, i2 = crf_pairs(410, 40, 40)
    alpha = ["N", "Aa", "Cc", "Gg", "Tt"]
    want = oracle.crf_beam_search_duplex(x1, i1, x2, i2, alpha, None, 5, 0.0, LSE | CR)
    assert fcd.crf_beam_search_duplex(x1, i1, x2, i2, alpha) == want  # character-reversal quirk included
    bad = band(40, 40, 6)
    bad[7, 0] = 35
    with pytest.raises(RuntimeError, match="Invalid envelope values"):
        fcd.crf_beam_search_duplex(x1, i1, x2, i2, "NACGT", bad)
    with pytest.raises(ValueError, match="beam_size cannot be 0"):
        fcd.crf_beam_search_duplex(x1, i1, x2, i2, "NACGT", None, 0)
    x1[:] = np.nan
    with pytest.raises(RuntimeError, match="Failed to compare values"):
        fcd.crf_beam_search_duplex(x1, i1, x2, i2, "NACGT")


def duplex_fuzz_seed(fcd, seed, mode):
    """Random shapes, beams, thresholds and random VALID envelopes (monotone, overlapping rows) as
    well as a few invalid ones: strings and error texts must equal the correctly-rounded oracle's."""
    if True:
        rng = np.random.default_rng(seed)
        N = int(rng.integers(3, 7))
        B = int(rng.integers(1, 4))
        T1, T2 = int(rng.integers(2, 60)), int(rng.integers(2, 60))
        beam = int(rng.choice([1, 2, 5, 9]))
        thr = float(rng.choice([0.0, 0.05, 0.15]))
        collapse = bool(rng.integers(0, 2))
        x1, x2 = pairs(seed, B, T1, T2, N)
        envs = None
        kind = int(rng.integers(0, 4))
        if kind >= 1:
            envs = np.zeros((B, T1, 2), np.uint64)
            for b in range(B):
                lo = np.sort(rng.integers(0, T2, size=T1))
                lo[0] = 0
                w = rng.integers(1, T2 + 1, size=T1)
                hi = np.minimum(T2, lo + w)
                hi = np.maximum.accumulate(hi)
                for i in range(1, T1):           # rows must overlap or touch: lo(i) <= hi(i-1)
                    lo[i] = min(lo[i], hi[i - 1])
                if kind == 3 and T1 > 2:         # sometimes break the envelope
                    lo[T1 // 2] = min(T2, hi[T1 // 2 - 1] + 1)
                    hi[T1 // 2] = min(T2, max(hi[T1 // 2], lo[T1 // 2]))
                envs[b, :, 0], envs[b, :, 1] = lo, hi
        alpha = "N" + "ACGTUV"[:N - 1]
        want = oracle_strings(x1, x2, alpha, envs, beam, thr, collapse, mode | CR)
        got = gpu_strings(fcd, x1, x2, alpha, envs, beam, thr, collapse, mode)
        assert got == want, (seed, N, B, T1, T2, beam, thr, collapse, kind)


def special_values_case(fcd, seed, mode):
    """A random banded pair batch with special posteriors injected -- exactly 1, exactly 0, above 1, NaN, in either read --
    vs the correctly-rounded oracle.  (NaN is where the ORDER of LogSpace::add's operands shows: max mode keeps one only
    as the first operand, so the merge of a node's candidates must fold in the reference's order.)"""
    rng = np.random.default_rng(seed)
    N = int(rng.integers(3, 7))
    B = int(rng.integers(1, 4))
    T1, T2 = int(rng.integers(20, 90)), int(rng.integers(20, 90))
    beam = int(rng.choice([1, 3, 5, 9]))
    thr = float(rng.choice([0.0, 0.05, 0.15]))
    collapse = bool(rng.integers(0, 2))
    x1, x2 = pairs(seed, B, T1, T2, N)
    for x in (x1, x2):
        for _ in range(int(rng.integers(1, 6))):
            b, t = int(rng.integers(0, B)), int(rng.integers(0, x.shape[1]))
            kind = int(rng.integers(0, 4))
            c = int(rng.integers(0, N))
            if kind == 0:
                x[b, t, :] = 0.0
                x[b, t, c] = 1.0
            elif kind == 1:
                x[b, t, c] = 0.0
            elif kind == 2:
                x[b, t, c] = 1.0 + float(rng.random())
            else:
                x[b, t, c] = np.nan
    w = int(rng.integers(6, 40))
    envs = np.stack([band(T1, T2, w)] * B)
    alpha = "N" + "ACGTUV"[:N - 1]
    want = oracle_strings(x1, x2, alpha, envs, beam, thr, collapse, mode | CR)
    got = gpu_strings(fcd, x1, x2, alpha, envs, beam, thr, collapse, mode)
    return got == want


@pytest.mark.parametrize("mode", [LSE, MAX], ids=["logsumexp", "max"])
def test_duplex_special_values_fuzz(fcd, mode):
    # (100369, 101027: cases the first soak of tools/duplex_soak.py found -- a NaN repeat-stay merged into the blank item)
    for seed in list(range(7000, 7024)) + [100369, 101027, 101074, 101121]:
        assert special_values_case(fcd, seed, mode), (seed, mode)


@pytest.mark.parametrize("mode", [LSE, MAX], ids=["logsumexp", "max"])
def test_duplex_fuzz(fcd, mode):
    for seed in range(5000, 5016):
        duplex_fuzz_seed(fcd, seed, mode)


@pytest.mark.parametrize("mode", [LSE, MAX], ids=["logsumexp", "max"])
def test_duplex_overlapping_calls(fcd, mode):
    """fcd_set_overlap (include/fcd.h) with the 2-D searches: every internal stream has its own region of the log-space
    buffer and of the arena.  Five batches of different shapes in flight on three streams deliver what they deliver in
    stream order (and the first of them the oracle's strings)."""
    import torch
    from fast_ctc_decode_amd import _native as nat
    shapes = [(4, 120, 110, 16), (7, 90, 100, 12), (3, 150, 150, 20), (6, 60, 64, 8), (5, 130, 120, 16)]
    data = []
    for k, (B, T1, T2, w) in enumerate(shapes):
        x1, x2 = pairs(700 + 10 * mode + k, B, T1, T2)
        data.append((x1, x2, np.stack([band(T1, T2, w)] * B)))
    serial = [fcd.beam_search_duplex_batch_raw(x1, x2, e, 5, 0.1, True, logadd_mode=mode).cpu() for x1, x2, e in data]
    x1, x2, e = data[0]
    assert gpu_strings(fcd, x1, x2, "NACGT", e, 5, 0.1, True, mode) == oracle_strings(x1, x2, "NACGT", e, 5, 0.1, True, mode | CR)
    h = nat.default_handle()
    h.set_overlap(3)
    try:
        if torch.cuda.is_available():
            dev = [(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), c) for a, b, c in data]
            rs = [fcd.beam_search_duplex_batch_raw(a, b, c, 5, 0.1, True, logadd_mode=mode) for a, b, c in dev]
            outs = [r.cpu() for r in rs]
        else:  # (the emulator: host arrays, the staged call joins before it copies back)
            outs = [fcd.beam_search_duplex_batch_raw(a, b, c, 5, 0.1, True, logadd_mode=mode).cpu() for a, b, c in data]
        for got, want in zip(outs, serial):
            np.testing.assert_array_equal(got.status, want.status)
            np.testing.assert_array_equal(got.out_len, want.out_len)
            for i in range(len(got.out_len)):
                n = int(got.out_len[i])
                np.testing.assert_array_equal(got.labels[i, :n], want.labels[i, :n])
    finally:
        h.set_overlap(0)


def test_crf_duplex_device_tensors(fcd):
    """The zero-copy (torch ROCm tensor) entry of the CRF duplex search equals the host entry."""
    torch = pytest.importorskip("torch")
    ps = [crf_pairs(600 + i, 50, 48) for i in range(3)]
    X1 = np.stack([p[0] for p in ps]); I1 = np.stack([p[1] for p in ps])
    X2 = np.stack([p[2] for p in ps]); I2 = np.stack([p[3] for p in ps])
    env = np.broadcast_to(band(50, 48, 10), (3, 50, 2)).copy()
    host = fcd.crf_beam_search_duplex_batch_raw(X1, I1, X2, I2, env, 5, 0.0)
    dev = fcd.crf_beam_search_duplex_batch_raw(torch.from_numpy(X1).cuda(), I1, torch.from_numpy(X2).cuda(), I2,
                                               env, 5, 0.0).cpu()
    for i in range(3):
        assert int(dev.status[i]) == int(host.status[i]) == 0
        n = int(host.out_len[i])
        assert int(dev.out_len[i]) == n
        np.testing.assert_array_equal(dev.labels[i, :n], host.labels[i, :n])
        want = oracle.crf_beam_search_duplex(ps[i][0], ps[i][1], ps[i][2], ps[i][3], "NACGT", env[i], 5, 0.0,
                                             LSE | CR)
        assert "".join("NACGT"[l] for l in host.labels[i, :n]) == want


def crf_duplex_fuzz_seed(fcd, seed, mode):
    """Random CRF pairs (state counts, alphabets, beams, thresholds, envelopes, init scores with ties):
    consensus strings and error texts must equal the correctly rounded oracle's."""
    rng = np.random.default_rng(seed)
    S = int(rng.choice([1, 2, 4, 4, 5]))
    N = int(rng.integers(3, 6))
    T1, T2 = int(rng.integers(2, 45)), int(rng.integers(2, 45))
    beam = int(rng.choice([1, 3, 5, 8]))
    thr = float(rng.choice([0.0, 0.0, 0.05, 0.15]))
    x1 = rng.random((T1, S, N), dtype=np.float32)
    x2 = rng.random((T2, S, N), dtype=np.float32)
    if rng.integers(0, 3) == 0:
        x1 = (rng.integers(1, 4, size=x1.shape) / 4.0).astype(np.float32)   # ties
        x2 = (rng.integers(1, 4, size=x2.shape) / 4.0).astype(np.float32)
    x1 /= x1.sum(-1, keepdims=True)
    x2 /= x2.sum(-1, keepdims=True)
    i1 = np.round(rng.random(S) * 2).astype(np.float32) / 2
    i2 = np.round(rng.random(S) * 2).astype(np.float32) / 2
    env = None
    if rng.integers(0, 2):
        w = int(rng.integers(3, 20))
        i = np.arange(T1)
        c = (i * T2) // T1
        env = np.stack([np.maximum(0, c - w), np.minimum(T2, c + w + 1)], 1).astype(np.uint64)
        env[0, 0] = 0
        env[-1, 1] = T2
        env[1:, 0] = np.minimum(env[1:, 0], env[:-1, 1])
    alpha = "N" + "ACGTUV"[:N - 1]
    omode = mode | CR
    try:
        want = oracle.crf_beam_search_duplex(x1, i1, x2, i2, alpha, env, beam, thr, omode)
    except RuntimeError as e:
        want = str(e)
    try:
        got = fcd.crf_beam_search_duplex(x1.astype(np.float32), i1, x2.astype(np.float32), i2, alpha, env, beam, thr,
                                         logadd_mode=(0 if mode == LSE else 1))
    except RuntimeError as e:
        got = str(e)
    if "panic" in want:
        assert "abort" in got, (seed, S, N, T1, T2, beam, thr, got, want)
    else:
        assert got == want, (seed, S, N, T1, T2, beam, thr)


@pytest.mark.parametrize("mode", [LSE, MAX], ids=["logsumexp", "max"])
def test_crf_duplex_fuzz(fcd, mode):
    for seed in range(8000, 8012):
        crf_duplex_fuzz_seed(fcd, seed, mode)


@pytest.mark.parametrize("mode", [LSE, MAX], ids=["logsumexp", "max"])
def test_duplex_tie_counters(fcd, mode):
    """fcd_result.ambiguous through the duplex kernels == the oracle's per-search counters
    (fcdo_duplex_last_ambiguous), pair by pair -- on generator data, on quantised data that forces exact ties,
    with beam 8 (up to 40 candidates: the > 20 case) and beam 3 (never above 20), on failing pairs too."""
    rng = np.random.default_rng(17 + mode)
    T = 140
    x1, x2 = pairs(900 + mode, 8, T, T)
    # pairs 4..7: probabilities quantised to 1/8 steps -- equal candidates at nearly every step
    for x in (x1, x2):
        q = np.round(rng.random((4, T, 5)) * 8) / 8 + 0.125
        x[4:] = (q / np.linalg.norm(q, axis=2, keepdims=True)).astype(np.float32)
    # pairs 4, 5: the four symbols share one probability in every row -- sibling candidates tie exactly in both
    # log-add modes (same parent, same factors, identical windows over read 2)
    for x in (x1, x2):
        b = rng.random((2, T, 1)) * 0.5 + 0.2
        sym = np.broadcast_to(np.sqrt((1 - b * b) / 4), (2, T, 4))
        x[4:6] = np.concatenate([b, sym], 2).astype(np.float32)
    x1[7, 50:] = 0.0  # runs out of beam half way
    envs = np.stack([band(T, T, 20)] * 8)
    seen = np.zeros(2, np.int64)
    thr = 0.02
    for beam in (8, 3):
        r = fcd.beam_search_duplex_batch_raw(x1, x2, envs, beam, thr, True, logadd_mode=mode,
                                             count_ambiguous=True).cpu()
        plain = fcd.beam_search_duplex_batch_raw(x1, x2, envs, beam, thr, True, logadd_mode=mode).cpu()
        assert np.array_equal(r.status, plain.status) and np.array_equal(r.out_len, plain.out_len)
        for i in range(8):
            try:
                oracle.beam_search_duplex(x1[i], x2[i], "NACGT", envs[i], beam, thr, True, mode | CR)
                st = 0
            except RuntimeError:
                st = 1
            assert (int(r.status[i]) != 0) == (st != 0)
            want = oracle.duplex_last_ambiguous()
            assert tuple(int(v) for v in r.ambiguous[i]) == want, (beam, i, r.ambiguous[i], want)
            seen += np.array(want)
            if beam == 3:
                assert want[0] == 0  # 15 candidates at most: never the pdqsort case
    assert seen[0] > 0 and seen[1] > 0  # the test data really exercises both counters


def test_logadd_fast_paths_exhaustive_on_device(fcd):
    """Every f32 argument of both fast-path domains of LogSpace::add, on the device build itself (whose ln_1p takes
    the hardware reciprocal instead of the IEEE division the host verifier sees): wherever Ziv's test trusts the fast
    binary64 value, its f32 rounding equals the library routine's.  1.1e9 + 2.0e8 arguments, 0 mismatches."""
    import ctypes as C
    import struct

    torch = pytest.importorskip("torch")
    from fast_ctc_decode_amd import _native as nat
    h = nat.default_handle()
    bits = lambda f: struct.unpack("<I", struct.pack("<f", f))[0]
    for which, lo, hi in ((0, 0x80000000, bits(-86.0)), (1, bits(2.0 ** -126), bits(1.0))):
        counts = torch.zeros(3, dtype=torch.int64, device="cuda")
        h.check(h.lib.fcd_logadd_sweep_dev(h.ptr, which, lo, hi, C.c_void_p(counts.data_ptr())))
        h.synchronize()
        n, slow, bad = (int(v) for v in counts.cpu())
        assert n == hi - lo + 1 and bad == 0, (which, n, slow, bad)
        assert slow < n * 1e-5          # the slow path stays rare (r01 host sweep: 517 / 397 arguments)
        print("device sweep", "exp" if which == 0 else "ln_1p", n, "arguments,", slow, "to the slow path, 0 wrong")


def _libm():
    import ctypes as C
    m = C.CDLL("libm.so.6")
    for f in ("expf", "logf", "log1pf"):
        getattr(m, f).restype = C.c_float
        getattr(m, f).argtypes = [C.c_float]
    return m


def _libm_apply(name, x):
    """f(x) with the C library's binary32 routine, element by element through the oracle's helper (a C loop)"""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    oracle.lib.fcdo_libm_apply({"expf": 0, "logf": 1, "log1pf": 2}[name], x.ctypes.data, out.ctypes.data, x.size)
    return out


needs_glibc235 = pytest.mark.skipif(__import__("platform").libc_ver() != ("glibc", "2.35"),
                                    reason="FCD_LOGADD_LOGSUMEXP_GLIBC235 reproduces glibc 2.35; this host links another libm")


@needs_glibc235
def test_glibc235_device_functions_equal_the_hosts_libm(fcd):
    """csrc/glibc235_math.h as compiled for gfx950 (v_fma_f64, IEEE division) against the libm of this host -- glibc 2.35
    on the build image and the GPU box -- on 12.6 M arguments per function: every 1021st binary32 pattern, and dense
    stretches where the routines change branch.  (All 2^32 arguments: tools/verify/verify_glibc235.c on the host,
    tools/verify/verify_glibc235_gpu.py on the device; profiles/r04_glibc235_verify.txt.)"""
    torch = pytest.importorskip("torch")
    from fast_ctc_decode_amd import _native as nat
    h = nat.default_handle()
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    pats = np.arange(0, 1 << 32, 1021, dtype=np.uint64).astype(np.uint32)
    dense = [np.arange(c - (1 << 20), c + (1 << 20), dtype=np.int64).astype(np.uint32) for c in
             (0x3f800000, 0x3ed413d7, 0xc2b00000, 0x42b00000, 0x00800000, 0xc2cff1b4)]
    x = np.concatenate([pats] + dense).view(np.float32)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    try:
        for which, name in enumerate(("expf", "logf", "log1pf")):
            h.check(h.lib.fcd_debug_glibc235_dev(h.ptr, which, xd.data_ptr(), yd.data_ptr(), x.size))
            torch.cuda.synchronize()
            got = yd.cpu().numpy()
            with np.errstate(all="ignore"):
                want = _libm_apply(name, x)
            same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
            assert same.all(), (name, int((~same).sum()), x[~same][:4], got[~same][:4], want[~same][:4])
    finally:
        h.reset_stream()


@needs_glibc235
def test_duplex_glibc235_mode_equals_the_oracle_on_the_hosts_libm(fcd):
    """FCD_LOGADD_LOGSUMEXP_GLIBC235: strings IDENTICAL to the oracle on the host's libm -- the arithmetic the reference
    computes on this machine -- on every pair (the correctly-rounded default flavour is held to >= 90 % against it:
    test_duplex_vs_host_libm_many_pairs), plain and CRF, banded and default envelopes."""
    x1, x2 = pairs(8100, 48, 140, 130)
    envs = np.stack([band(140, 130, 20)] * 48)
    got = gpu_strings(fcd, x1, x2, "NACGT", envs, 5, 0.1, True, "logsumexp_glibc235")
    assert got == oracle_strings(x1, x2, "NACGT", envs, 5, 0.1, True, LSE)
    y1, y2 = pairs(8101, 6, 50, 60)
    got = gpu_strings(fcd, y1, y2, "NACGT", None, 5, 0.0, False, "logsumexp_glibc235")
    assert got == oracle_strings(y1, y2, "NACGT", None, 5, 0.0, False, LSE)
    cx1, ci1, cx2, ci2 = crf_pairs(8102, 90, 84)
    cenv = band(90, 84, 16)
    assert fcd.crf_beam_search_duplex(cx1, ci1, cx2, ci2, "NACGT", cenv, 5, 0.0, logadd_mode="logsumexp_glibc235") == \
        oracle.crf_beam_search_duplex(cx1, ci1, cx2, ci2, "NACGT", cenv, 5, 0.0, LSE)
