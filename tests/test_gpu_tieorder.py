"""the order of EQUAL probabilities in the prune (src/search.rs:122,262, src/duplex.rs:620,807).
Both orders are selectable (include/fcd.h: FCD_TIE_PDQ178 -- the default, Rust 1.78's sort_unstable_by --
and FCD_TIE_STABLE); under either, every kernel family must equal the oracle under the same order, bit for bit, on
inputs built to tie; the two orders must really differ on them (else the test says nothing); and the reads of the
BASELINE configurations whose result round 3 found to depend on the order come out as the restatement says."""
import numpy as np
import pytest

import test_gpu_duplex as D
import test_gpu_parity as P
from oracle import oracle
from test_pdq178 import check_against_oracle, coop_lists, device_coop_sort, device_sort, tie_lists
from tie_util import ORDERS, tie_order

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fcd():
    import fast_ctc_decode_amd as m
    return m


def quantised(seed, B, T, N, levels=4):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, levels, size=(B, T, N)) / float(levels)).astype(np.float32)


def test_device_routine_equals_the_oracle_restatement(fcd):
    """csrc/pdq178.h as compiled for gfx950, lists in LDS (and, above 2048 elements, in HBM)"""
    torch = pytest.importorskip("torch")
    from fast_ctc_decode_amd import _native as nat

    class Dev:
        def __init__(self, a):
            self.t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()
            self.ptr = self.t.data_ptr()

    lists = list(tie_lists())
    h = nat.default_handle(0)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        out, lens = device_sort(nat.load(), h, lists, Dev,
                                lambda d, shape, dt: d.t.cpu().numpy().view(dt).reshape(shape))
    finally:
        h.reset_stream()
    assert check_against_oracle(out, lens, lists) > 100


@pytest.mark.parametrize("planes", [1, 3, 5, 8])
def test_cooperative_routine_equals_the_oracle_restatement(fcd, planes):
    """csrc/pdq178_wave.h as compiled for gfx950: the whole wavefront replaying the quicksort on one list"""
    torch = pytest.importorskip("torch")
    from fast_ctc_decode_amd import _native as nat

    class Dev:
        def __init__(self, a):
            self.t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()
            self.ptr = self.t.data_ptr()

    lists = coop_lists(planes)
    h = nat.default_handle(0)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        out, lens = device_coop_sort(nat.load(), h, lists, planes, Dev,
                                     lambda d, shape, dt: d.t.cpu().numpy().view(dt).reshape(shape))
    finally:
        h.reset_stream()
    assert check_against_oracle(out, lens, lists) > (10 if planes == 1 else 100)
    if planes in (3, 5):  # the searches only need the kept prefix: segments behind it are dropped
        h.set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            for keep in (1, 5, 32):
                out, lens = device_coop_sort(nat.load(), h, lists, planes, Dev,
                                             lambda d, shape, dt: d.t.cpu().numpy().view(dt).reshape(shape), keep=keep)
                check_against_oracle(out, lens, lists, keep=keep)
        finally:
            h.reset_stream()


def results(fcd, x, beam, thr, collapse, kernel, lengths=None):
    r = fcd.beam_search_batch_raw(x, beam, thr, collapse, lengths=lengths, kernel=kernel)
    return [(int(r.status[i]), r.labels[i, :int(r.out_len[i])].tolist(), r.path[i, :int(r.out_len[i])].tolist())
            for i in range(x.shape[0])]


CASES = [  # (N, beam, kernels): every family, every candidate-count regime above 20
    (5, 5, (0, 1, 2, 3, 4)),     # 25 candidates: the headline shape (two reads per wavefront), one read, lane, generic
    (7, 8, (0, 1, 3, 4)),        # 56: one read per wavefront, groups of eight
    (5, 12, (0, 1, 2, 4)),       # 60: groups of five
    (5, 32, (0, 1, 4)),          # 160: lane kernel, two reads per wavefront; choose_pivot's median of medians
    (8, 64, (1, 4)),             # 512: lane kernel, one read per wavefront; partition_in_blocks with full blocks
    (4, 5, (1, 2, 3, 4)),        # 20 candidates: never above 20 -- both orders must agree
]


@pytest.mark.parametrize("N,beam,kernels", CASES)
def test_both_tie_orders_every_kernel(fcd, N, beam, kernels):
    x = quantised(900 + N + beam, 6, 160, N)
    x[:, :, 0] = np.maximum(x[:, :, 0], 0.25)
    lengths = np.array([160, 159, 1, 0, 100, 33], np.int64)
    got = {}
    for order in ORDERS:
        with tie_order(fcd, order):
            for k in kernels:
                P.check_beam(fcd, x, beam, 0.0, kernel=k)
                P.check_beam(fcd, x, beam, 0.1, False, lengths=lengths, kernel=k)
            got[order] = results(fcd, x, beam, 0.0, True, kernels[0])
    differ = sum(a != b for a, b in zip(got["pdq178"], got["stable"]))
    assert (differ > 0) == (beam * N > 20), differ


def test_tie_order_is_per_handle_too(fcd):
    """fcd_set_tie_order on a handle overrides the process default (and FCD_TIE_DEFAULT gives it back)"""
    from fast_ctc_decode_amd import _native as nat
    x = quantised(7, 4, 150, 5)
    h = nat.default_handle(0)
    with tie_order(fcd, "pdq178"):
        a = results(fcd, x, 5, 0.0, True, 0)
        h.set_tie_order(nat.TIE_STABLE)
        try:
            b = results(fcd, x, 5, 0.0, True, 0)
        finally:
            h.set_tie_order(nat.TIE_DEFAULT)
        assert results(fcd, x, 5, 0.0, True, 0) == a
    with tie_order(fcd, "stable"):
        assert results(fcd, x, 5, 0.0, True, 0) == b
    assert a != b


@pytest.mark.parametrize("order", ORDERS)
def test_crf_beam_both_orders(fcd, order):
    """search::crf_beam_search (:122): quantised transition scores, S = 4 and 16, wave / lane / generic"""
    rng = np.random.default_rng(31)
    for S, beam, kernels in ((4, 5, (0, 1, 2, 3, 4)), (16, 12, (0, 1, 4)), (4, 32, (1, 4))):
        x = (rng.integers(1, 5, size=(4, 120, S, 5)) / 4.0).astype(np.float32)
        init = np.zeros((4, S), np.float32)
        init[np.arange(4), rng.integers(0, S, 4)] = 1.0
        with tie_order(fcd, order):
            want = [oracle.crf_beam_search(x[i], init[i], "NACGT", beam, 0.0) for i in range(4)]
            for k in kernels:
                r = fcd.crf_beam_search_batch_raw(x, init, beam, 0.0, kernel=k).cpu()
                for i in range(4):
                    n = int(r.out_len[i])
                    assert int(r.status[i]) == 0
                    got = ("".join("NACGT"[l] for l in r.labels[i, :n]), r.path[i, :n].tolist())
                    assert got == want[i], (S, beam, k, i)


@pytest.mark.parametrize("mode", [D.LSE, D.MAX], ids=["logsumexp", "max"])
def test_duplex_both_orders(fcd, mode):
    """duplex::beam_search / crf_beam_search (src/duplex.rs:620,807): beam 8 x 5 symbols = 40 candidates; in max mode
    (the reference's default build) equal probabilities are the rule, not the exception"""
    x1, x2 = D.pairs(520 + mode, 6, 140, 130)
    envs = np.stack([D.band(140, 130, 20)] * 6)
    q1, q2 = quantised(521, 4, 90, 5), quantised(522, 4, 90, 5)
    q1[:, :, 0] = np.maximum(q1[:, :, 0], 0.25)
    q2[:, :, 0] = np.maximum(q2[:, :, 0], 0.25)
    qenv = np.stack([D.band(90, 90, 12)] * 4)
    got = {}
    for order in ORDERS:
        with tie_order(fcd, order):
            a = D.gpu_strings(fcd, x1, x2, "NACGT", envs, 8, 0.05, True, mode)
            assert a == D.oracle_strings(x1, x2, "NACGT", envs, 8, 0.05, True, mode | D.CR)
            b = D.gpu_strings(fcd, q1, q2, "NACGT", qenv, 8, 0.0, True, mode)
            assert b == D.oracle_strings(q1, q2, "NACGT", qenv, 8, 0.0, True, mode | D.CR)
            cx1, ci1, cx2, ci2 = D.crf_pairs(523, 80, 76)
            cenv = D.band(80, 76, 16)
            c = fcd.crf_beam_search_duplex(cx1, ci1, cx2, ci2, "NACGT", cenv, 8, 0.0, logadd_mode=mode)
            assert c == oracle.crf_beam_search_duplex(cx1, ci1, cx2, ci2, "NACGT", cenv, 8, 0.0, mode | D.CR)
            got[order] = (a, b, c)
    if mode == D.MAX:
        assert got["pdq178"] != got["stable"]


def test_baseline_reads_that_depend_on_the_order(fcd):
    """Round 3 (profiles/r03a_pdqsort_ties.jsonl): of BASELINE config 2's 4096 reads exactly 1198 and 3588, and of the
    first 1024 reads of config 3's shard exactly 173, 257 and 914, decode differently under the two orders.  The
    kernels now follow either: equal to the oracle under both, different from each other on exactly those reads
    (their neighbours, decoded in the same launch, do not change)."""
    import bench
    for seed, n, beam, reads, kernels in ((1, 4096, 5, [1197, 1198, 3588, 3589], (0, 1, 3, 4)),
                                          (2, 1024, 32, [173, 174, 257, 914], (0, 1))):
        x = np.ascontiguousarray(bench.make_batch(seed, n)[reads])
        got = {}
        for order in ORDERS:
            with tie_order(fcd, order):
                for k in kernels:
                    P.check_beam(fcd, x, beam, 0.1, kernel=k)
                got[order] = results(fcd, x, beam, 0.1, True, 0)
        differ = [reads[i] for i in range(len(reads)) if got["pdq178"][i] != got["stable"][i]]
        assert differ == [r for r in reads if r in (1198, 3588, 173, 257, 914)], differ


def test_std_forms_of_the_replay_on_the_gpu(fcd):
    """FCD_PDQ178_STD_FORM / fcd_debug_set_pdq178_std_form on the MI355X: under form 3 (the earlier forms of the two
    routines std changed in 2023 -- what a compiled rustc-1.65 std does, tests/test_rust165_pdqsort.py) the serial and the
    wave / register routines give the permutations the vector file lists for it, and whole searches equal the oracle
    under the same form; back at form 0 everything is as before."""
    import json
    import os
    torch = pytest.importorskip("torch")
    from fast_ctc_decode_amd import _native as nat

    class Dev:
        def __init__(self, a):
            self.t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()
            self.ptr = self.t.data_ptr()

    back = lambda d, shape, dt: d.t.cpu().numpy().view(dt).reshape(shape)  # noqa: E731
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "verify", "pdq178_vectors.json")
    cases = [c for c in json.load(open(path))["cases"] if "perm_gp" in c]
    lists = [np.array(c["bits"], np.uint32).view(np.float32) for c in cases]
    rng = np.random.default_rng(31)
    x = (rng.integers(0, 4, size=(6, 200, 5)) / 4.0).astype(np.float32)
    x[:, :, 0] = np.maximum(x[:, :, 0], 0.25)
    lib = nat.load()
    h = nat.default_handle(0)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for form, key in ((3, "perm_gp"), (1, "perm_g"), (0, "perm")):
            h.set_pdq178_std_form(form)
            perms = [np.array(c.get(key, c["perm"]), np.int64) for c in cases]
            out, lens = device_sort(lib, h, lists, Dev, back)
            for i, perm in enumerate(perms):
                assert np.array_equal((out[i, :lens[i]] & np.uint64(0xFFFFFFFF)).astype(np.int64), perm), (form, i)
            out, lens = device_coop_sort(lib, h, lists, 8, Dev, back)
            for i, perm in enumerate(perms):
                assert np.array_equal((out[i, :lens[i]] & np.uint64(0xFFFFFFFF)).astype(np.int64), perm), (form, i)
            with oracle.pdq_std_form(form):
                for beam, kernels in ((12, (0, 1, 3)), (32, (1, 4))):
                    for k in kernels:
                        P.check_beam(fcd, x, beam, 0.0, kernel=k)
    finally:
        h.set_pdq178_std_form(0)
        h.reset_stream()
