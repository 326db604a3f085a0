"""The C oracle's CRF / duplex halves against an independent, deliberately naive Python restatement
(tests/naive_reference.py, written from src/search.rs:38-157 and src/duplex.rs without looking at the oracle):
identical strings / paths / errors on random small cases -- wobbly and invalid envelopes, beams wider than the
node count, nodes that leave and re-enter the beam, S in {4, 16}, both log-add modes.  CPU only.
tools/naive_crosscheck.py runs the same generators for as many cases as wanted."""
import numpy as np
import pytest

import naive_reference as naive
from kat_cases import reference_style_rows
from oracle import oracle


@pytest.fixture(autouse=True)
def _stable_tie_order():
    """The naive restatement sorts with Python's (stable) sort: this cross-check is about the CRF / duplex logic, so
    the oracle runs under the stable tie rule here (its default follows Rust 1.78's pdqsort since round 4; that order
    has its own cross-check, tests/test_pdq178.py)."""
    with oracle.unstable_sort("stable"):
        yield


def rows(rng, T, N, peaky):
    if peaky:
        z = rng.normal(size=(T, N)).astype(np.float32) * 3.0
        e = np.exp(z - z.max(-1, keepdims=True))
        return (e / e.sum(-1, keepdims=True)).astype(np.float32)
    return reference_style_rows(rng, T, N)


def wobbly_envelope(rng, T1, T2, kind):
    """(T1, 2) uint64: kind 0 sliding band, 1 bounds that jump / stall / move back (still valid), 2 default (full),
    3 possibly invalid (gaps, empty rows)"""
    if kind == 2 or T2 < 3:
        return np.stack([np.zeros(T1, np.uint64), np.full(T1, T2, np.uint64)], 1)
    w = int(rng.integers(2, 7))
    centre = np.linspace(0, T2 - 1, T1)
    lo = np.clip(centre - w, 0, T2 - 1).astype(np.int64)
    hi = np.clip(centre + w + 1, 1, T2).astype(np.int64)
    if kind in (1, 3):
        lo = np.clip(lo + rng.integers(-2, 3, T1), 0, T2 - 1)
        hi = np.clip(hi + rng.integers(-2, 4, T1), 1, T2)
    if kind != 3:
        lo[0] = 0
        hi = np.maximum(hi, lo + 1)
        for i in range(1, T1):                  # lo(i) <= hi(i-1): the reference's validity rule
            lo[i] = min(lo[i], max(hi[i - 1], hi[:i].max()))
        hi = np.maximum(hi, lo + 1)
    return np.stack([lo, hi], 1).astype(np.uint64)


def duplex_case(seed):
    rng = np.random.default_rng(seed)
    T1, T2 = int(rng.integers(1, 34)), int(rng.integers(1, 34))
    N = int(rng.choice([3, 5]))
    x1, x2 = rows(rng, T1, N, seed % 3 == 0), rows(rng, T2, N, seed % 5 == 0)
    if seed % 11 == 0:
        x1[rng.integers(0, T1)] = 0.0           # a dead row: ln 0 = -inf everywhere
    env = wobbly_envelope(rng, T1, T2, seed % 4)
    beam = int(rng.choice([1, 2, 3, 5, 8]))
    thr = float(rng.choice([0.0, 0.05, 0.1]))
    return x1, x2, env, "NACGT"[:N], beam, thr, bool(seed % 7 != 0)


def crf_case(seed):
    rng = np.random.default_rng(seed)
    S = int(rng.choice([4, 16]))
    T1, T2 = int(rng.integers(1, 22)), int(rng.integers(1, 22))
    x1 = rng.random((T1, S, 5), dtype=np.float32) + np.float32(0.01)
    x2 = rng.random((T2, S, 5), dtype=np.float32) + np.float32(0.01)
    i1, i2 = rng.random(S, dtype=np.float32), rng.random(S, dtype=np.float32)
    env = wobbly_envelope(rng, T1, T2, seed % 4)
    beam = int(rng.choice([1, 2, 4, 6]))
    thr = float(rng.choice([0.0, 0.1, 0.15]))
    return x1, i1, x2, i2, env, beam, thr


def naive_outcome(fn):
    try:
        return fn()
    except naive.SearchError as e:
        return "error: " + str(e)
    except (AssertionError, IndexError):
        return "error: panic"


def oracle_outcome(fn):
    try:
        return fn()
    except RuntimeError as e:
        msg = str(e)
        return "error: panic" if "panic" in msg else "error: " + msg


def check_duplex(seed, max_mode):
    x1, x2, env, alpha, beam, thr, collapse = duplex_case(seed)
    mode = oracle.MAXMODE if max_mode else (oracle.LOGSUMEXP | oracle.MATH_CR)
    want = oracle_outcome(lambda: oracle.beam_search_duplex(x1, x2, alpha, env, beam, thr, collapse, mode))
    got = naive_outcome(lambda: naive.Duplex(max_mode).beam_search(
        x1.tolist(), x2.tolist(), alpha, env.tolist(), beam, thr, collapse))
    assert got == want, (seed, max_mode, got, want)
    return want


def check_crf_duplex(seed, max_mode):
    x1, i1, x2, i2, env, beam, thr = crf_case(seed)
    mode = oracle.MAXMODE if max_mode else (oracle.LOGSUMEXP | oracle.MATH_CR)
    want = oracle_outcome(lambda: oracle.crf_beam_search_duplex(x1, i1, x2, i2, "NACGT", env, beam, thr, mode))
    got = naive_outcome(lambda: naive.Duplex(max_mode).crf_beam_search(
        x1.tolist(), i1.tolist(), x2.tolist(), i2.tolist(), "NACGT", env.tolist(), beam, thr))
    assert got == want, (seed, max_mode, got, want)
    return want


def inject_specials(rng, *arrays):
    """A few special posteriors -- NaN, exactly 1 (a row's whole mass), exactly 0, above 1, +inf -- at random places."""
    for _ in range(int(rng.integers(1, 5))):
        a = arrays[int(rng.integers(0, len(arrays)))]
        idx = tuple(int(rng.integers(0, n)) for n in a.shape)
        kind = int(rng.integers(0, 5))
        if kind == 0:
            a[idx] = np.nan
        elif kind == 1:
            a[idx[:-1]] = 0.0
            a[idx] = 1.0
        elif kind == 2:
            a[idx] = 0.0
        elif kind == 3:
            a[idx] = 1.0 + float(rng.random())
        else:
            a[idx] = np.inf


def check_duplex_special(seed, max_mode):
    x1, x2, env, alpha, beam, thr, collapse = duplex_case(seed)
    inject_specials(np.random.default_rng(seed + 99), x1, x2)
    mode = oracle.MAXMODE if max_mode else (oracle.LOGSUMEXP | oracle.MATH_CR)
    want = oracle_outcome(lambda: oracle.beam_search_duplex(x1, x2, alpha, env, beam, thr, collapse, mode))
    got = naive_outcome(lambda: naive.Duplex(max_mode).beam_search(
        x1.tolist(), x2.tolist(), alpha, env.tolist(), beam, thr, collapse))
    assert got == want, (seed, max_mode, got, want)
    return want


def check_crf_duplex_special(seed, max_mode):
    x1, i1, x2, i2, env, beam, thr = crf_case(seed)
    inject_specials(np.random.default_rng(seed + 99), x1, x2)
    mode = oracle.MAXMODE if max_mode else (oracle.LOGSUMEXP | oracle.MATH_CR)
    want = oracle_outcome(lambda: oracle.crf_beam_search_duplex(x1, i1, x2, i2, "NACGT", env, beam, thr, mode))
    got = naive_outcome(lambda: naive.Duplex(max_mode).crf_beam_search(
        x1.tolist(), i1.tolist(), x2.tolist(), i2.tolist(), "NACGT", env.tolist(), beam, thr))
    assert got == want, (seed, max_mode, got, want)
    return want


def check_crf_1d(seed):
    rng = np.random.default_rng(seed)
    S = int(rng.choice([4, 16]))
    T = int(rng.integers(1, 60))
    x = rng.random((T, S, 5), dtype=np.float32)
    init = rng.random(S, dtype=np.float32)
    beam = int(rng.choice([1, 3, 5, 9]))
    thr = float(rng.choice([0.0, 0.1, 0.3, 0.6]))
    alpha = ["N", "Ab", "C", "G", "T"] if seed % 4 == 0 else "NACGT"
    want = oracle_outcome(lambda: oracle.crf_beam_search(x, init, alpha, beam, thr))
    got = naive_outcome(lambda: naive.crf_beam_search(x.tolist(), init.tolist(), alpha, beam, thr))
    assert got == want, (seed, got, want)
    return want


@pytest.mark.parametrize("max_mode", [False, True], ids=["logsumexp", "max"])
def test_duplex_oracle_equals_the_naive_restatement(max_mode):
    outcomes = [check_duplex(seed, max_mode) for seed in range(400)]
    errors = sum(o.startswith("error") for o in outcomes)
    assert 0 < errors < len(outcomes) // 2      # invalid envelopes and dead rows occur, most cases decode
    assert len({o for o in outcomes if not o.startswith("error")}) > 150


@pytest.mark.parametrize("max_mode", [False, True], ids=["logsumexp", "max"])
def test_crf_duplex_oracle_equals_the_naive_restatement(max_mode):
    outcomes = [check_crf_duplex(seed, max_mode) for seed in range(1000, 1250)]
    assert sum(not o.startswith("error") for o in outcomes) > 120


def test_crf_beam_search_oracle_equals_the_naive_restatement():
    outcomes = [check_crf_1d(seed) for seed in range(2000, 2400)]
    assert sum(not (isinstance(o, str) and o.startswith("error")) for o in outcomes) > 250


@pytest.mark.parametrize("max_mode", [False, True], ids=["logsumexp", "max"])
def test_duplex_special_posteriors_oracle_equals_the_naive_restatement(max_mode):
    """NaN / 1 / 0 / > 1 / +inf posteriors: where the ORDER of LogSpace::add's operands shows (max mode keeps a NaN only as
    its first operand) -- the oracle, which the GPU soaks are measured against, and the independent restatement agree."""
    outcomes = [check_duplex_special(seed, max_mode) for seed in range(5000, 5300)]
    outcomes += [check_crf_duplex_special(seed, max_mode) for seed in range(6000, 6150)]
    assert sum(not o.startswith("error") for o in outcomes) > 150


def check_beam_1d(seed, special):
    """search::beam_search: the oracle against the naive restatement (sequence AND path), optionally with special
    posteriors injected"""
    rng = np.random.default_rng(seed)
    N = int(rng.integers(2, 8))
    T = int(rng.integers(1, 120))
    style = int(rng.integers(0, 3))
    if style == 0:
        x = reference_style_rows(rng, T, N)
    elif style == 1:
        x = (rng.integers(0, 4, size=(T, N)) / 4.0).astype(np.float32)
    else:
        x = rows(rng, T, N, True)
    if special:
        inject_specials(np.random.default_rng(seed + 99), x)
    beam = int(rng.choice([1, 2, 3, 5, 8, 16]))   # (<= 20 candidates per step up to beam 4; above: see DESIGN.md section 2)
    thr = float(rng.choice([0.0, 0.01, 0.1, 0.3])) * (1.0 / N) / 0.34   # (the wrapper demands thr < 1 / N)
    thr = min(thr, 0.99 / N)
    collapse = bool(rng.integers(0, 2))
    alpha = "NACGTUVW"[:N]
    want = oracle_outcome(lambda: oracle.beam_search(x, alpha, beam, thr, collapse))
    got = naive_outcome(lambda: naive.beam_search(x.tolist(), alpha, beam, thr, collapse))
    if not isinstance(want, str):
        want = (want[0], [int(v) for v in want[1]])
    if not isinstance(got, str):
        got = (got[0], [int(v) for v in got[1]])
    assert got == want, (seed, special, got, want)
    return want


@pytest.mark.parametrize("special", [False, True], ids=["ordinary", "special-posteriors"])
def test_beam_search_oracle_equals_the_naive_restatement(special):
    outcomes = [check_beam_1d(seed, special) for seed in range(7000, 7500)]
    assert sum(not isinstance(o, str) for o in outcomes) > (100 if special else 300)
