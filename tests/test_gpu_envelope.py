"""The duplex alignment-band estimator (csrc/envelope.hip) against its executable specification
(tests/envelope_model.py) -- integer work, so the comparison is exact -- and the properties that make
an envelope usable by duplex::beam_search (src/duplex.rs:485-488)."""
import numpy as np
import pytest

import envelope_model as em
from kat_cases import reference_style_rows
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fcd():
    import fast_ctc_decode_amd as m
    return m


def warped_pair(rng, n_labels, N=5, noise=0.02, dwell=(1, 6), mutate=0.0):
    """Two posterior matrices rendering (almost) the same label sequence with independent dwell times:
    what a duplex pair looks like.  Returns x1, x2 and the label times of both renderings."""
    seq = rng.integers(1, N, n_labels)
    seq2 = seq.copy()
    if mutate:
        flip = rng.random(n_labels) < mutate
        seq2[flip] = rng.integers(1, N, int(flip.sum()))

    def render(s):
        rows, times = [], []
        for l in s:
            for _ in range(int(rng.integers(0, 3))):           # blanks between labels
                rows.append(0)
            times.append(len(rows))
            for _ in range(int(rng.integers(dwell[0], dwell[1]))):
                rows.append(int(l))
            rows.append(0)
        rows = np.asarray(rows)
        x = np.full((len(rows), N), noise, np.float32)
        x[np.arange(len(rows)), rows] = 1.0 - noise * (N - 1)
        return x, np.asarray(times)

    x1, t1 = render(seq)
    x2, t2 = render(seq2)
    return x1, x2, t1, t2


def check_valid(env, T1, T2):
    lo, hi = env[:T1, 0].astype(np.int64), env[:T1, 1].astype(np.int64)
    assert lo[0] == 0 and hi[-1] == T2
    assert (lo < hi).all() and (hi <= T2).all()
    assert (lo[1:] <= hi[:-1]).all()                   # consecutive rows touch (:485-488)
    assert (np.diff(lo) >= 0).all() and (np.diff(hi) >= 0).all()


def model_env(fcd, x1, x2, band):
    r1 = fcd.viterbi_search_batch_raw(x1[None], True)
    r2 = fcd.viterbi_search_batch_raw(x2[None], True)
    n1, n2 = int(r1.out_len[0]), int(r2.out_len[0])
    return em.envelope(r1.labels[0, :n1], r1.path[0, :n1], x1.shape[0],
                       r2.labels[0, :n2], r2.path[0, :n2], x2.shape[0], band)


@pytest.mark.parametrize("seed", range(12))
def test_envelope_equals_model(fcd, seed):
    rng = np.random.default_rng(9000 + seed)
    style = seed % 4
    if style == 0:    # clean warped copies
        x1, x2, _, _ = warped_pair(rng, int(rng.integers(1, 120)))
    elif style == 1:  # substitutions between the two reads
        x1, x2, _, _ = warped_pair(rng, int(rng.integers(5, 150)), mutate=0.15)
    elif style == 2:  # unrelated random matrices (the alignment has little to hold on to)
        x1 = reference_style_rows(rng, int(rng.integers(1, 300)), 5)
        x2 = reference_style_rows(rng, int(rng.integers(1, 300)), 5)
    else:             # columns straddling 64-column chunk boundaries of the DP
        x1, x2, _, _ = warped_pair(rng, 64 + int(rng.integers(0, 3)), dwell=(1, 2))
    band = int(rng.choice([0, 1, 4, 16, 64]))
    want = model_env(fcd, x1, x2, band)
    got = fcd.estimate_envelope(x1, x2, band)
    np.testing.assert_array_equal(got, want)
    check_valid(got, x1.shape[0], x2.shape[0])


def test_envelope_batch_ragged_and_device(fcd):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(77)
    pairs = [warped_pair(rng, n, mutate=0.05)[:2] for n in (3, 40, 90, 1, 64)]
    T1 = max(p[0].shape[0] for p in pairs)
    T2 = max(p[1].shape[0] for p in pairs)
    B = len(pairs)
    X1 = np.zeros((B, T1, 5), np.float32)
    X2 = np.zeros((B, T2, 5), np.float32)
    l1 = np.array([p[0].shape[0] for p in pairs], np.int64)
    l2 = np.array([p[1].shape[0] for p in pairs], np.int64)
    for i, (a, b) in enumerate(pairs):
        X1[i, :l1[i]], X2[i, :l2[i]] = a, b
    host = fcd.estimate_envelope_batch(X1, X2, 8, l1, l2)
    dev = fcd.estimate_envelope_batch(torch.from_numpy(X1).cuda(), torch.from_numpy(X2).cuda(), 8, l1, l2)
    dev = dev.cpu().numpy().view(np.uint64)
    for i, (a, b) in enumerate(pairs):
        want = model_env(fcd, a, b, 8)
        np.testing.assert_array_equal(host[i, :l1[i]], want)
        np.testing.assert_array_equal(dev[i, :l1[i]], want)
        check_valid(want, int(l1[i]), int(l2[i]))


def test_envelope_contains_true_alignment_and_decodes_like_full(fcd):
    """On warped copies the band must contain the true label-to-label alignment, and decoding inside
    it must give the same consensus as the reference's default (full) envelope."""
    rng = np.random.default_rng(4242)
    for _ in range(4):
        x1, x2, t1, t2 = warped_pair(rng, 60)
        env = fcd.estimate_envelope(x1, x2, 12)
        lo, hi = env[t1, 0].astype(np.int64), env[t1, 1].astype(np.int64)
        assert ((lo <= t2) & (t2 < hi)).all()
        full = fcd.beam_search_duplex(x1, x2, "NACGT", None, 5, 0.1)
        banded = fcd.beam_search_duplex(x1, x2, "NACGT", env, 5, 0.1)
        assert banded == full
        # and the oracle agrees on the banded search (the envelope is just an argument to it)
        assert banded == oracle.beam_search_duplex(x1, x2, "NACGT", env, 5, 0.1, True,
                                                   oracle.LOGSUMEXP | oracle.MATH_CR)


def test_envelope_limits_and_edges(fcd):
    rng = np.random.default_rng(5)
    x = reference_style_rows(rng, 10, 5)
    assert fcd.estimate_envelope(np.zeros((0, 5), np.float32), x).shape == (0, 2)
    e = fcd.estimate_envelope(x, x, 0)        # identical reads, zero band: the diagonal itself
    check_valid(e, 10, 10)
    # limits follow the number of LABELS, not the time axis: 40000 blank rows are fine ...
    big = np.zeros((1, 40000, 5), np.float32)
    big[..., 0] = 1.0
    check_valid(fcd.estimate_envelope_batch(big, big[:, :30000], 8)[0], 40000, 30000)
    # ... 20000 labels in read 2 are not
    alt = np.zeros((1, 40000, 5), np.float32)
    alt[0, 0::2, 1] = 1.0
    alt[0, 1::2, 2] = 1.0
    with pytest.raises(RuntimeError, match="envelope estimator"):
        fcd.estimate_envelope_batch(alt, alt, 8)
