"""csrc/pdq178.h -- the device routine the beam kernels run on their tie-flagged steps (FCD_TIE_PDQ178) -- against
the oracle's restatement of Rust 1.78's sort_unstable_by (oracle/fcd_oracle.c, DEFINE_PDQSORT).  The two were written
separately (explicit stack vs recursion, u64 elements vs structs); they must produce the same PERMUTATION, equal keys
included, on every list.  Here through tests/hipemu (no GPU); tests/test_gpu_tieorder.py runs the same lists on the
MI355X."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from test_pdqsort_restatement import _patterns


def orderable(p):
    """device_utils.h make_key's upper word: a total order on non-NaN f32, larger = greater probability"""
    u = (np.asarray(p, np.float32) + np.float32(0)).view(np.uint32).astype(np.uint64)
    neg = (u & 0x80000000) != 0
    return np.where(neg, u ^ 0xFFFFFFFF, u ^ 0x80000000).astype(np.uint64)


def tie_lists(seed=0):
    """(prob, tag) lists: adversarial patterns at the lengths where pdqsort changes gear (20 / 21, 50, 128-element
    blocks, 2 * 128 + ...), plus what the searches really produce: few distinct values among 21..512 candidates."""
    rng = np.random.default_rng(seed)
    for n in list(range(0, 70)) + [100, 127, 128, 129, 160, 255, 256, 257, 258, 320, 511, 512, 700, 2000, 2500]:
        for p in _patterns(rng, n):
            yield np.ascontiguousarray(p, np.float32)
    for _ in range(400):
        n = int(rng.integers(21, 513))
        k = int(rng.integers(1, 12))
        vals = rng.random(k, dtype=np.float32)
        p = vals[rng.integers(0, k, n)]
        if rng.random() < 0.5:  # mostly distinct with a few repeated values, like a beam step
            q = rng.random(n, dtype=np.float32)
            m = rng.random(n) < 0.3
            p = np.where(m, p, q).astype(np.float32)
        yield np.ascontiguousarray(p, np.float32)


def device_sort(lib, handle, lists, to_dev, from_dev):
    stride = max(1, max(len(p) for p in lists))
    buf = np.zeros((len(lists), stride), np.uint64)
    lens = np.zeros(len(lists), np.int32)
    for i, p in enumerate(lists):
        buf[i, :len(p)] = (orderable(p) << np.uint64(32)) | np.arange(len(p), dtype=np.uint64)
        lens[i] = len(p)
    d_buf, d_lens = to_dev(buf), to_dev(lens)
    rc = lib.fcd_debug_pdq178_sort_dev(handle.ptr, d_buf.ptr, len(lists), stride, d_lens.ptr)
    assert rc == 0, lib.fcd_last_error(handle.ptr)
    handle.synchronize()
    return from_dev(d_buf, buf.shape, np.uint64), lens


def check_against_oracle(out, lens, lists, keep=None):
    differs = 0
    for i, p in enumerate(lists):
        n = int(lens[i])
        got = (out[i, :n] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        _, want = oracle.pdqsort_desc(p, np.arange(n, dtype=np.int32))
        if keep is not None:  # only a prefix was asked for; the rest must still be the same elements
            assert np.array_equal(np.sort(got), np.sort(want)), (i, n)
            got, want = got[:keep], want[:keep]
        assert np.array_equal(got, want), (i, n)
        differs += int(not np.array_equal(got, np.argsort(-p, kind="stable")))
    return differs


def coop_lists(planes, seed=1):
    """lists for the wave-cooperative routine (at most 64 * planes positions each): every regime of the serial routine,
    plus beam-like lists with few distinct values"""
    cap = 64 * planes
    rng = np.random.default_rng(seed)
    out = []
    pool = [p for p in tie_lists(seed) if len(p) <= cap]
    for i, p in enumerate(pool):
        room = cap - len(p)
        partner = pool[(i * 7 + 3) % len(pool)]
        out.append(p)
        out.append(partner if len(partner) <= room else partner[:room])
    for _ in range(300):  # what the kernels hand over: two lists of up to half the positions each
        for _side in range(2):
            n = int(rng.integers(0, cap // 2 + 1))
            k = int(rng.integers(1, 9))
            vals = rng.random(k, dtype=np.float32)
            p = vals[rng.integers(0, k, n)] if n else np.zeros(0, np.float32)
            if rng.random() < 0.6 and n:
                p = np.where(rng.random(n) < 0.35, p, rng.random(n, dtype=np.float32)).astype(np.float32)
            out.append(np.ascontiguousarray(p, np.float32))
    return out


def device_coop_sort(lib, handle, lists, planes, to_dev, from_dev, keep=1 << 20):
    stride = max(1, max(len(p) for p in lists))
    buf = np.zeros((len(lists), stride), np.uint64)
    lens = np.zeros(len(lists), np.int32)
    for i, p in enumerate(lists):
        buf[i, :len(p)] = (orderable(p) << np.uint64(32)) | np.arange(len(p), dtype=np.uint64)
        lens[i] = len(p)
    d_buf, d_lens = to_dev(buf), to_dev(lens)
    rc = lib.fcd_debug_pdq178_coop_sort_dev(handle.ptr, d_buf.ptr, len(lists), stride, d_lens.ptr, planes, keep)
    assert rc == 0, lib.fcd_last_error(handle.ptr)
    handle.synchronize()
    return from_dev(d_buf, buf.shape, np.uint64), lens


class _HostBuf:
    def __init__(self, a):
        self.a = np.ascontiguousarray(a).copy()
        self.ptr = self.a.ctypes.data


def test_device_routine_equals_the_oracle_restatement_emulated():
    from emu_util import emulated_kernels
    from fast_ctc_decode_amd import _native as nat
    lists = list(tie_lists())
    with emulated_kernels() as lib:
        h = nat.default_handle(0)
        # the emulator's "device" memory is host memory
        out, lens = device_sort(lib, h, lists, _HostBuf, lambda d, shape, dt: d.a.reshape(shape))
    assert check_against_oracle(out, lens, lists) > 100  # and it really is another order than the stable one


@pytest.mark.parametrize("planes", [1, 3, 5, 8])
def test_cooperative_routine_equals_the_oracle_restatement_emulated(planes):
    """csrc/pdq178_wave.h -- the whole wavefront replaying the quicksort on a list (one wave-uniform segment at a time,
    ballots for partition_in_blocks' offsets, leaves ranked in parallel) -- must produce the serial routine's permutation"""
    from emu_util import emulated_kernels
    from fast_ctc_decode_amd import _native as nat
    lists = coop_lists(planes)
    with emulated_kernels() as lib:
        h = nat.default_handle(0)
        out, lens = device_coop_sort(lib, h, lists, planes, _HostBuf, lambda d, shape, dt: d.a.reshape(shape))
    assert check_against_oracle(out, lens, lists) > (10 if planes == 1 else 100)
    if planes in (3, 5):  # the searches only need the kept prefix: segments behind it are dropped
        with emulated_kernels() as lib:
            h = nat.default_handle(0)
            for keep in (1, 5, 32):
                out, lens = device_coop_sort(lib, h, lists, planes, _HostBuf, lambda d, shape, dt: d.a.reshape(shape), keep=keep)
                check_against_oracle(out, lens, lists, keep=keep)


def test_tie_order_api_emulated():
    from emu_util import emulated_kernels
    from fast_ctc_decode_amd import _native as nat
    with emulated_kernels() as lib:
        h = nat.default_handle(0)
        assert h.tie_order() == nat.TIE_PDQ178          # the process default
        h.set_tie_order(nat.TIE_STABLE)
        assert h.tie_order() == nat.TIE_STABLE
        h.set_tie_order(nat.TIE_DEFAULT)
        assert lib.fcd_set_default_tie_order(nat.TIE_STABLE) == 0 and h.tie_order() == nat.TIE_STABLE
        assert lib.fcd_set_default_tie_order(nat.TIE_PDQ178) == 0 and h.tie_order() == nat.TIE_PDQ178
        assert lib.fcd_set_default_tie_order(7) != 0 and lib.fcd_set_tie_order(h.ptr, 7) != 0


def test_committed_vectors_match_every_restatement_emulated():
    """tools/verify/pdq178_vectors.json is what a holder of rustc 1.78.0 checks with tools/verify/pdq178_check.rs: it has
    to say what the oracle, csrc/pdq178.h and csrc/pdq178_wave.h really do -- all three, on every list it holds."""
    import json
    import os
    from emu_util import emulated_kernels
    from fast_ctc_decode_amd import _native as nat
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "verify", "pdq178_vectors.json")
    doc = json.load(open(path))
    assert doc["meta"]["cases"] == len(doc["cases"]) >= 1500
    lists = [np.array(c["bits"], np.uint32).view(np.float32) for c in doc["cases"]]
    perms = [np.array(c["perm"], np.int64) for c in doc["cases"]]
    assert min(len(p) for p in lists) >= 21 and max(len(p) for p in lists) <= 512
    for p, perm in zip(lists, perms):
        _, want = oracle.pdqsort_desc(p, np.arange(len(p), dtype=np.int32))
        assert np.array_equal(want, perm)
    with emulated_kernels() as lib:
        h = nat.default_handle(0)
        out, lens = device_sort(lib, h, lists, _HostBuf, lambda d, shape, dt: d.a.reshape(shape))
        for i, perm in enumerate(perms):
            assert np.array_equal((out[i, :lens[i]] & np.uint64(0xFFFFFFFF)).astype(np.int64), perm), i
        out, lens = device_coop_sort(lib, h, lists, 8, _HostBuf, lambda d, shape, dt: d.a.reshape(shape))
        for i, perm in enumerate(perms):
            assert np.array_equal((out[i, :lens[i]] & np.uint64(0xFFFFFFFF)).astype(np.int64), perm), i


@pytest.mark.parametrize("form,key", [(1, "perm_g"), (2, "perm_p"), (3, "perm_gp")])
def test_std_forms_of_the_device_routines_emulated(form, key):
    """FCD_PDQ178_STD_FORM / fcd_debug_set_pdq178_std_form: the EARLIER forms of the two routines std changed in 2023
    (csrc/pdq178.h g_std_form) -- the serial routine and the wave / register ones must give the permutations the vector
    file lists for that form (form 3 is what a compiled rustc-1.65 std produces: tests/test_rust165_pdqsort.py)"""
    import json
    import os
    from emu_util import emulated_kernels
    from fast_ctc_decode_amd import _native as nat
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "verify", "pdq178_vectors.json")
    cases = [c for c in json.load(open(path))["cases"] if key in c or len(c["bits"]) in (50, 130, 257)]
    assert sum(key in c for c in cases) > 50
    lists = [np.array(c["bits"], np.uint32).view(np.float32) for c in cases]
    perms = [np.array(c.get(key, c["perm"]), np.int64) for c in cases]
    back = lambda d, shape, dt: d.a.reshape(shape)  # noqa: E731
    with emulated_kernels() as lib:
        h = nat.default_handle(0)
        assert lib.fcd_debug_get_pdq178_std_form() == 0
        h.set_pdq178_std_form(form)
        try:
            assert lib.fcd_debug_get_pdq178_std_form() == form
            out, lens = device_sort(lib, h, lists, _HostBuf, back)
            for i, perm in enumerate(perms):
                assert np.array_equal((out[i, :lens[i]] & np.uint64(0xFFFFFFFF)).astype(np.int64), perm), i
            out, lens = device_coop_sort(lib, h, lists, 8, _HostBuf, back)
            for i, perm in enumerate(perms):
                assert np.array_equal((out[i, :lens[i]] & np.uint64(0xFFFFFFFF)).astype(np.int64), perm), i
            short = [(p, perm) for p, perm in zip(lists, perms) if len(p) <= 64]
            out, lens = device_coop_sort(lib, h, [p for p, _ in short], 1, _HostBuf, back)  # (registers only)
            for i, (_, perm) in enumerate(short):
                assert np.array_equal((out[i, :lens[i]] & np.uint64(0xFFFFFFFF)).astype(np.int64), perm), i
            assert lib.fcd_debug_set_pdq178_std_form(h.ptr, 4) != 0
        finally:
            h.set_pdq178_std_form(0)
        out, lens = device_sort(lib, h, lists[:40], _HostBuf, back)  # back to the default form
        for i, c in enumerate(cases[:40]):
            assert np.array_equal((out[i, :lens[i]] & np.uint64(0xFFFFFFFF)).astype(np.int64), np.array(c["perm"], np.int64)), i


def test_searches_follow_the_std_form_emulated():
    """a whole search under std form 3 equals the oracle under the same form, on every kernel family (reads built to tie)"""
    from emu_util import emulated_kernels
    from fast_ctc_decode_amd import _native as nat
    import fast_ctc_decode_amd as fcd
    import test_gpu_parity as P
    rng = np.random.default_rng(31)
    sets = []
    for N, beam, kernels in ((5, 12, (0, 1, 3)), (5, 32, (1, 4)), (7, 8, (3,))):
        x = (rng.integers(0, 4, size=(3, 120, N)) / 4.0).astype(np.float32)
        x[:, :, 0] = np.maximum(x[:, :, 0], 0.25)
        sets.append((x, beam, kernels))

    def decoded(x, beam):
        out = oracle.batch_outputs(x.shape[0], x.shape[1])
        lab, path, lens, st = oracle.beam_search_batch(x, beam, 0.0, True, 1, out=out)
        return [(int(st[i]), lab[i, :lens[i]].tolist(), path[i, :lens[i]].tolist()) for i in range(x.shape[0])]

    # (the reads really depend on the form: else the test below would say nothing)
    plain = [decoded(x, beam) for x, beam, _ in sets]
    with oracle.pdq_std_form(3):
        earlier = [decoded(x, beam) for x, beam, _ in sets]
    assert sum(a != b for u, v in zip(plain, earlier) for a, b in zip(u, v)) >= 3
    with emulated_kernels():
        h = nat.default_handle(0)
        h.set_pdq178_std_form(3)
        try:
            with oracle.pdq_std_form(3):
                for x, beam, kernels in sets:
                    for k in kernels:
                        P.check_beam(fcd, x, beam, 0.0, kernel=k)
        finally:
            h.set_pdq178_std_form(0)
