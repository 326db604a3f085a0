"""Pins the CPU oracle (oracle/fcd_oracle.c) to every known-answer test the reference holds
(SURVEY.md section 4, K1-K18).  CPU only."""
import ctypes as C

import numpy as np
import pytest

import kat_cases
from oracle import oracle


_COMPILED = {}


def _compiled_recurse():
    """address of the rustc-1.65 core::slice::sort::recurse libcst's native module carries (tools/verify/rust165_pdqsort.py),
    or None"""
    if "addr" not in _COMPILED:
        import os
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "verify"))
        import rust165_pdqsort as R
        s = R.Rust165Sort()
        ok = bool(s.path) and "897e37553bba" in R.rustc_commit(s.path) and s.select()
        _COMPILED["addr"] = s.address_of_ascending_24() if ok else None
        _COMPILED["keep"] = s
    return _COMPILED["addr"]


@pytest.fixture(autouse=True, params=["stable", "pdqsort", "rustc165"])
def _tie_order_of_the_unstable_sort(request):
    """Every KAT must hold under every order of EQUAL probabilities the oracle can impose above 20 candidates: the stable
    rule, its restatement of Rust 1.78's pdqsort, and -- where the image has it -- Rust's OWN quicksort, compiled by rustc
    1.65, sorting in the restatement's place.  The reference's vectors do not depend on that order."""
    if request.param == "rustc165":
        addr = _compiled_recurse()
        if addr is None:
            pytest.skip("no compiled core::slice::sort::recurse here")
        with oracle.unstable_sort("pdqsort"), oracle.external_recurse(addr):
            yield
        return
    with oracle.unstable_sort(request.param):
        yield


@pytest.mark.parametrize("case", kat_cases.ONE_D_CASES + kat_cases.CRF_CASES,
                         ids=lambda f: f.__name__)
def test_kat_1d(case):
    case(oracle)


class _Mode:
    """The duplex KATs must hold in both builds of the reference (fastexp on / off)."""

    def __init__(self, mode):
        self.mode = mode

    def beam_search_duplex(self, *a, **k):
        return oracle.beam_search_duplex(*a, logadd_mode=self.mode, **k)


@pytest.mark.parametrize("mode", [oracle.LOGSUMEXP, oracle.MAXMODE], ids=["logsumexp", "max"])
@pytest.mark.parametrize("case", kat_cases.DUPLEX_CASES, ids=lambda f: f.__name__)
def test_kat_duplex(case, mode):
    case(_Mode(mode))


def test_api_shape():
    kat_cases.api_beam_search(oracle)
    kat_cases.api_viterbi_search(oracle)


def test_k5_phred():  # src/search.rs:513-525
    f32 = np.float32
    probs = [f32(0.0), f32(0.5), f32(1.0) - f32(1e-1), f32(1.0) - f32(1e-2), f32(1.0) - f32(1e-3),
             f32(1.0) - f32(1e-4), f32(1.0) - f32(1e-5), f32(1.0) - f32(1e-6), f32(1.0)]
    assert [oracle.phred(float(p)) for p in probs] == list("!$+5?IIII")


def test_k17_tree():  # src/tree.rs:200-269
    lib = oracle.lib
    t = lib.fcdo_tree_new(2)
    try:
        def walk(node):
            out = []
            while node >= 0:
                out.append((lib.fcdo_tree_label(t, node), lib.fcdo_tree_data(t, node)))
                node = lib.fcdo_tree_parent(t, node)
            return out

        assert lib.fcdo_tree_label(t, -1) == -1
        assert lib.fcdo_tree_get_child(t, -1, 0) == -1
        assert lib.fcdo_tree_get_child(t, -1, 1) == -1
        assert lib.fcdo_tree_add_node(t, -1, 1, 10) == 0
        assert lib.fcdo_tree_get_child(t, -1, 0) == -1
        assert lib.fcdo_tree_get_child(t, -1, 1) == 0
        assert lib.fcdo_tree_label(t, 0) == 1 and lib.fcdo_tree_parent(t, 0) == -1
        assert lib.fcdo_tree_get_child(t, 0, 0) == -1 and lib.fcdo_tree_get_child(t, 0, 1) == -1
        assert lib.fcdo_tree_add_node(t, 0, 0, 20) == 1
        assert lib.fcdo_tree_add_node(t, 0, 1, 30) == 2
        assert lib.fcdo_tree_add_node(t, -1, 0, 40) == 3
        assert lib.fcdo_tree_add_node(t, 2, 1, 50) == 4
        assert lib.fcdo_tree_get_child(t, 0, 0) == 1 and lib.fcdo_tree_get_child(t, 0, 1) == 2
        assert lib.fcdo_tree_get_child(t, -1, 0) == 3 and lib.fcdo_tree_get_child(t, 2, 1) == 4
        assert lib.fcdo_tree_get_child(t, 2, 0) == -1
        assert lib.fcdo_tree_len(t) == 5
        assert walk(4) == [(1, 50), (1, 30), (1, 10)]
        assert walk(1) == [(0, 20), (1, 10)]
        assert walk(3) == [(0, 40)]
    finally:
        lib.fcdo_tree_free(t)


NEG_INF = np.float32(-np.inf)


def _pairs(items):
    """items: list of ('g'|'l', prob) -> (len,2) f32 array of (label, gap) log-probs."""
    a = np.full((len(items), 2), NEG_INF, np.float32)
    for i, (kind, p) in enumerate(items):
        a[i, 1 if kind == "g" else 0] = np.log(np.float32(p), dtype=np.float32)
    return a


def _get_gap(pairs, offset, at):
    lab, gap = C.c_float(0), C.c_float(0)
    p = pairs if len(pairs) else np.zeros((1, 2), np.float32)
    oracle.lib.fcdo_secondary_get(p.ctypes.data_as(C.c_void_p), len(pairs), offset, at,
                                  C.byref(lab), C.byref(gap))
    return np.float32(gap.value)


def test_k18_secondary_get():  # src/duplex.rs:841-889
    p = _pairs([("g", 0.1), ("g", 0.2), ("g", 0.3)])
    g = p[:, 1]
    assert [_get_gap(p, 0, i) for i in (-1, 0, 1, 2, 3)] == [NEG_INF, g[0], g[1], g[2], NEG_INF]
    assert [_get_gap(p, 3, i) for i in (-1, 0, 2, 3, 4, 5, 6)] == [
        NEG_INF, NEG_INF, NEG_INF, g[0], g[1], g[2], NEG_INF]
    assert [_get_gap(p, -1, i) for i in (-2, -1, 0, 1, 2)] == [NEG_INF, g[0], g[1], g[2], NEG_INF]
    e = p[:0]
    for off, ats in ((0, (-1, 0, 1)), (-1, (-2, -1, 0)), (4, (3, 4, 5))):
        assert all(_get_gap(e, off, a) == NEG_INF for a in ats)


def _upd(pairs, offset, lo, hi, mode=0):
    p = pairs if len(pairs) else np.zeros((1, 2), np.float32)
    return np.float32(oracle.lib.fcdo_secondary_update_max(p.ctypes.data_as(C.c_void_p),
                                                           len(pairs), offset, lo, hi, mode))


@pytest.mark.parametrize("mode", [0, 1])
def test_k18_update_max_empty(mode):  # src/duplex.rs:891-919
    e = _pairs([])
    imin, imax = -2**63, 2**63 - 1
    for lo, hi in ((0, 0), (-1, 0), (0, 1), (-1, 1), (imin, imax)):
        assert _upd(e, 0, lo, hi, mode) == NEG_INF


@pytest.mark.parametrize("mode", [0, 1])
def test_k18_update_max_values(mode):  # src/duplex.rs:921-993
    p = _pairs([("g", 0.1), ("l", 0.3), ("l", 0.2), ("l", 0.4), ("g", 0.5)])
    ln = lambda v: np.log(np.float32(v), dtype=np.float32)
    expect = [((0, 0), None), ((0, 2), None), ((0, 3), 0.1), ((2, 2), None), ((2, 3), 0.1),
              ((2, 4), 0.3), ((2, 5), 0.3), ((2, 6), 0.4), ((2, 7), 0.5), ((6, 7), 0.5),
              ((7, 7), None), ((2, 10), 0.5), ((3, 10), 0.5), ((8, 10), None)]
    for (lo, hi), v in expect:
        want = NEG_INF if v is None else ln(v)
        assert _upd(p, 2, lo, hi, mode) == want, (lo, hi)


def test_logspace_add():  # src/duplex.rs:42-63
    add = oracle.lib.fcdo_logspace_add
    ninf = float("-inf")
    assert add(ninf, ninf, 0) == ninf
    assert add(-1.0, ninf, 0) == -1.0 and add(ninf, -1.0, 0) == -1.0
    assert add(-1.0, -2.0, 0) == add(-2.0, -1.0, 0)
    assert abs(add(np.log(0.25), np.log(0.5), 0) - np.log(0.75)) < 1e-6
    assert add(-1.0, -2.0, 1) == -1.0  # max mode
    assert np.isnan(add(float("nan"), -1.0, 0)) and np.isnan(add(-1.0, float("nan"), 0))


def test_duplex_tie_statistics_instrument():
    """fcdo_duplex_tie_steps (analysis instrument, tools/duplex_ties.py): counts pruning steps and reports ties
    where the inputs force them -- two identical symbol columns make the two extensions of every entry tie."""
    from oracle import oracle
    rng = np.random.default_rng(9)
    T = 30
    x = rng.random((T, 5)).astype(np.float32)
    x[:, 2] = x[:, 1]                      # labels 1 and 2 always have equal probability
    x /= x.sum(-1, keepdims=True)
    env = np.stack([np.zeros(T, np.uint64), np.full(T, T, np.uint64)], 1)
    oracle.duplex_tie_steps(reset=True)
    oracle.beam_search_duplex(x, x, "NACGT", env, 5, 0.0, True, oracle.MAXMODE)
    st = oracle.duplex_tie_steps(reset=True)
    assert st["steps"] == T and st["boundary_tie"] + st["gt20_kept_tie"] > 0, st
    assert oracle.duplex_tie_steps()["steps"] == 0
