"""SURVEY.md 8a A4 -- the restatement of std's pattern-defeating quicksort against a REAL rustc-compiled one.

libcst's native module (rustc 1.65.0) carries core::slice::sort::recurse with its symbols; tools/verify/rust165_pdqsort.py
calls it on (key, node) records.  With the earlier forms of the two routines std changed in 2023 selected
(fcdo_set_pdq_std_form(3)) the oracle's restatement must equal it element for element; under the default forms (Rust 1.78
as recalled: what the kernels follow) every difference must sit on a list that reaches one of those two routines.
Skipped where the module (or binutils' nm) is missing: nothing but this test depends on it."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "verify"))
import rust165_pdqsort as R  # noqa: E402
from oracle import oracle  # noqa: E402


@pytest.fixture(scope="module")
def compiled():
    s = R.Rust165Sort()
    if not s.path:
        pytest.skip("libcst's native module is not installed")
    if "897e37553bba" not in R.rustc_commit(s.path):
        pytest.skip("libcst's native module was not built with rustc 1.65.0 (commit %s)" % R.rustc_commit(s.path))
    if not s.select():
        pytest.skip("no callable core::slice::sort::recurse found in %s" % s.path)
    return s


def test_the_second_compiled_instance_agrees_too():
    """libcst carries two monomorphisations (24-byte elements ascending by their first word, 16-byte ones DESCENDING by
    their second): two separately compiled copies of the routine, one answer"""
    s = R.Rust165Sort()
    if not s.path or "897e37553bba" not in R.rustc_commit(s.path) or not s.select(1):
        pytest.skip("no second callable core::slice::sort::recurse")
    first = R.Rust165Sort()
    assert first.select(0) and first.symbol != s.symbol
    rep = R.compare(s)
    assert rep["ok"] and rep["differ"][3] == [], rep["lines"]


def test_committed_vectors_pin_the_restatement_to_a_compiled_std(compiled):
    rep = R.compare(compiled)
    assert rep["ok"], rep["lines"]
    assert rep["differ"][3] == []
    # the two changed routines are really exercised by the vectors (else the statement above says little)
    assert len(rep["differ"][1]) > 20 and len(rep["differ"][2]) > 100


def test_alternative_permutations_in_the_vector_file_are_the_oracles():
    doc = json.load(open(os.path.join(ROOT, "tools", "verify", "pdq178_vectors.json")))
    seen = {"perm_g": 0, "perm_p": 0, "perm_gp": 0}
    for c in doc["cases"][::7]:
        p = np.array(c["bits"], np.uint32).view(np.float32)
        for key, form in (("perm_g", 1), ("perm_p", 2), ("perm_gp", 3)):
            with oracle.unstable_sort("pdqsort"), oracle.pdq_std_form(form):
                _, perm = oracle.pdqsort_desc(p, np.arange(len(p), dtype=np.int32))
            want = c.get(key, c["perm"])
            assert perm.tolist() == want
            seen[key] += key in c
    assert all(v > 0 for v in seen.values()), seen
    assert oracle.lib.fcdo_get_pdq_std_form() == 0


def test_random_lists_against_the_compiled_std(compiled):
    rng = np.random.default_rng(165)
    n_lists = reached = 0
    with oracle.unstable_sort("pdqsort"), oracle.pdq_std_form(3):
        for _ in range(4000):
            n = int(rng.integers(2, 400))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                p = rng.random(n, dtype=np.float32)
            elif kind == 1:
                k = int(rng.integers(1, 12))
                p = rng.random(k, dtype=np.float32)[rng.integers(0, k, n)]
            elif kind == 2:
                p = np.sort(rng.random(n, dtype=np.float32))
                if rng.random() < 0.5:
                    p = p[::-1].copy()
                for _ in range(int(rng.integers(0, 6))):
                    i, j = rng.integers(0, n, 2)
                    p[i], p[j] = p[j], p[i]
            else:
                p = (np.round(rng.random(n) * 16) / 16).astype(np.float32)
            p = np.ascontiguousarray(p, np.float32)
            real = compiled.sort(R.keys_of(p), np.arange(n, dtype=np.int64))
            oracle.pdq_path_counts(reset=True)
            _, perm = oracle.pdqsort_desc(p, np.arange(n, dtype=np.int32))
            b, q = oracle.pdq_path_counts()
            reached += (b > 0) or (q > 0)
            n_lists += 1
            assert np.array_equal(real, perm), (n, kind)
    assert reached > 200


def test_whole_searches_on_the_compiled_quicksort(compiled):
    """The oracle's beam searches with RUST'S OWN quicksort in place of the restatement (fcdo_set_external_recurse: the
    compiled rustc-1.65 routine sorts every list above 20 candidates) decode every read exactly as the oracle does under
    the earlier std forms -- plain, CRF-free 1-D searches at three beams and the duplex search -- and not as it does under
    the later forms on some of them (the inputs are built to tie)."""
    addr = compiled.address_of_ascending_24()
    if addr is None:
        pytest.skip("no instance over 24-byte ascending records")
    rng = np.random.default_rng(165)

    def decoded(x, beam):
        out = oracle.batch_outputs(x.shape[0], x.shape[1])
        lab, path, lens, st = oracle.beam_search_batch(x, beam, 0.0, True, 1, out=out)
        return [(int(st[i]), lab[i, :lens[i]].tolist(), path[i, :lens[i]].tolist()) for i in range(x.shape[0])]

    changed = 0
    for N, beam, B, T in ((5, 12, 6, 150), (5, 32, 6, 150), (7, 8, 6, 150), (8, 64, 2, 100)):
        x = (rng.integers(0, 4, size=(B, T, N)) / 4.0).astype(np.float32)
        x[:, :, 0] = np.maximum(x[:, :, 0], 0.25)
        with oracle.unstable_sort("pdqsort"):
            with oracle.external_recurse(addr):
                real = decoded(x, beam)
            with oracle.pdq_std_form(3):
                earlier = decoded(x, beam)
            later = decoded(x, beam)
        assert real == earlier, (N, beam)
        changed += sum(a != b for a, b in zip(real, later))
    assert changed >= 3
    # the duplex search (its own element type and comparator through the same routine)
    T = 300
    x1 = rng.random((T, 5), dtype=np.float32)
    x2 = (np.round(rng.random((T, 5)) * 4) / 4).astype(np.float32) + np.float32(0.01)
    x1 /= np.linalg.norm(x1, axis=-1, keepdims=True)
    x2 /= np.linalg.norm(x2, axis=-1, keepdims=True)
    i = np.arange(T)
    env = np.stack([np.maximum(0, i - 20), np.minimum(T, i + 20)], 1).astype(np.uint64)
    with oracle.unstable_sort("pdqsort"):
        with oracle.external_recurse(addr):
            real = oracle.beam_search_duplex(x1, x2, "NACGT", env, 8, 0.0, True)
        with oracle.pdq_std_form(3):
            earlier = oracle.beam_search_duplex(x1, x2, "NACGT", env, 8, 0.0, True)
    assert real == earlier


def test_a_second_toolchain_version_agrees_where_the_image_has_one():
    """cryptography's Rust binding in the image's conda environment was built in 2021 (rustc commit a178d0322ce2): an
    older std still, the same pdqsort -- its compiled routine (comparator closure zero-sized: another call shape) must
    agree with the restatement under the earlier forms as well"""
    mods = R.other_modules()
    if not mods:
        pytest.skip("no other Rust-built module with a pre-2023 core::slice::sort::recurse here")
    s = R.Rust165Sort(mods[0])
    if not s.select():
        pytest.skip("no callable core::slice::sort::recurse in %s" % mods[0])
    rep = R.compare(s)
    assert rep["ok"] and rep["differ"][3] == [], rep["lines"]


def test_the_kernels_routines_against_the_compiled_std_directly_emulated(compiled):
    """csrc/pdq178.h (serial) and csrc/pdq178_wave.h + pdq178_reg.h (the whole wavefront; registers only up to 64
    elements) -- the code the GPU kernels replay ties with, here on the lockstep emulator under std form 3 -- against
    the compiled rustc-1.65 routine itself, without the oracle in between"""
    from emu_util import emulated_kernels
    from fast_ctc_decode_amd import _native as nat
    from test_pdq178 import _HostBuf, device_coop_sort, device_sort
    rng = np.random.default_rng(1650)
    lists = []
    for _ in range(260):
        n = int(rng.integers(21, 400))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            p = rng.random(n, dtype=np.float32)
        elif kind == 1:
            k = int(rng.integers(1, 9))
            p = rng.random(k, dtype=np.float32)[rng.integers(0, k, n)]
        elif kind == 2:
            p = np.sort(rng.random(n, dtype=np.float32))[::-1].copy()
            p = (np.round(p * 16) / 16).astype(np.float32)
            for _ in range(int(rng.integers(0, 5))):
                i, j = rng.integers(0, n, 2)
                p[i], p[j] = p[j], p[i]
        else:
            h = rng.random((n + 1) // 2, dtype=np.float32)
            p = np.repeat(h, 2)[:n][rng.permutation(n)]
        lists.append(np.ascontiguousarray(p, np.float32))
    want = [compiled.sort(R.keys_of(p), np.arange(len(p), dtype=np.int64)) for p in lists]
    back = lambda d, shape, dt: d.a.reshape(shape)  # noqa: E731
    with emulated_kernels() as lib:
        h = nat.default_handle(0)
        h.set_pdq178_std_form(3)
        try:
            out, lens = device_sort(lib, h, lists, _HostBuf, back)
            for i, w in enumerate(want):
                assert np.array_equal((out[i, :lens[i]] & np.uint64(0xFFFFFFFF)).astype(np.int64), w), i
            out, lens = device_coop_sort(lib, h, lists, 8, _HostBuf, back)
            for i, w in enumerate(want):
                assert np.array_equal((out[i, :lens[i]] & np.uint64(0xFFFFFFFF)).astype(np.int64), w), i
            short = [i for i, p in enumerate(lists) if len(p) <= 64]
            out, lens = device_coop_sort(lib, h, [lists[i] for i in short], 1, _HostBuf, back)
            for j, i in enumerate(short):
                assert np.array_equal((out[j, :lens[j]] & np.uint64(0xFFFFFFFF)).astype(np.int64), want[i]), i
        finally:
            h.set_pdq178_std_form(0)
