"""The oracle's restatement of Rust 1.78's pdqsort (oracle/fcd_oracle.c, DEFINE_PDQSORT; written from memory; no
Rust source or toolchain here -- tests/test_rust165_pdqsort.py pins it against a COMPILED rustc-1.65 std found in the
image, except for the 2023 forms of two routines).  What can be checked without that binary: every sort it
performs is a correct descending sort and a permutation on adversarial patterns (sorted, reversed, constant,
few distinct keys, organ pipes: the inputs that reach partition_equal, break_patterns, the reversal in
choose_pivot, partial_insertion_sort and heapsort); distinct keys give the unique answer; it is deterministic;
searches without a flagged tie are unchanged by it (the reference's KATs run under both orders in
tests/test_oracle_kat.py)."""
import numpy as np
import pytest

from kat_cases import reference_style_rows
from oracle import oracle


def _patterns(rng, n):
    yield rng.random(n, dtype=np.float32)
    yield np.sort(rng.random(n, dtype=np.float32))
    yield np.sort(rng.random(n, dtype=np.float32))[::-1].copy()
    yield (rng.integers(0, 4, n) / 4).astype(np.float32)
    yield np.full(n, 0.5, np.float32)
    a = np.arange(n, dtype=np.float32)
    yield np.minimum(a, a[::-1])                      # organ pipe
    yield np.where(np.arange(n) % 2 == 0, a, -a)      # saw
    p = rng.random(n, dtype=np.float32)
    p[::3] = p[0] if n else 0
    yield p
    yield np.concatenate([np.sort(rng.random(n // 2, dtype=np.float32)), rng.random(n - n // 2, dtype=np.float32)])


def test_pdqsort_restatement_sorts():
    rng = np.random.default_rng(0)
    differs_from_stable = 0
    for n in list(range(0, 70)) + [100, 127, 128, 129, 160, 255, 256, 257, 320, 700, 2000]:
        for p in _patterns(rng, n):
            ids = np.arange(n, dtype=np.int32)
            sp, sn = oracle.pdqsort_desc(p, ids)
            assert np.all(sp[:-1] >= sp[1:]), n
            assert np.array_equal(np.sort(sn), ids) and np.array_equal(p[sn], sp), n
            sp2, sn2 = oracle.pdqsort_desc(p, ids)
            assert np.array_equal(sn, sn2)  # deterministic (break_patterns' generator is seeded by the length)
            stable = np.argsort(-p, kind="stable")
            if len(set(p.tolist())) == n:
                assert np.array_equal(sn, stable)
            if n <= 20:
                assert np.array_equal(sn, stable)  # insertion sort: the pinned case
            differs_from_stable += int(not np.array_equal(sn, stable))
    assert differs_from_stable > 0  # it really is another order of equal keys


def test_searches_without_flagged_ties_are_unchanged():
    rng = np.random.default_rng(5)
    x = reference_style_rows(rng, 12 * 300, 5).reshape(12, 300, 5)
    for i in range(12):
        st, labels, path, amb = oracle.beam_search_ambiguous(x[i], 32, 0.1)
        with oracle.unstable_sort("pdqsort"):
            st2, labels2, path2, amb2 = oracle.beam_search_ambiguous(x[i], 32, 0.1)
        assert list(amb) == list(amb2)  # the counters are taken on the stably sorted list in both modes
        if amb[0] == 0 or amb[1] == 0:
            assert st == st2 and np.array_equal(labels, labels2) and np.array_equal(path, path2)
