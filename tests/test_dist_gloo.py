"""world_size-2 `gloo` test of the multi-GPU path on CPU: shard -> decode shard -> ONE gather.

The decode step itself needs a GPU, so here the shard decoder is the CPU oracle (test
infrastructure); everything else -- shard_bounds, pack/unpack, the single gather, global read
order -- is the product's fast_ctc_decode_amd.dist code, exactly as bench.py drives it over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_decode(x):
    from fast_ctc_decode_amd.api import BatchResult
    from oracle import oracle
    labels, path, lens, status = oracle.beam_search_batch(np.asarray(x), 5, 0.1, True, 1)
    return BatchResult(labels.astype(np.uint8), path.astype(np.uint32), lens.astype(np.uint32),
                       status.astype(np.int32))


def _worker(rank, world, port, n_reads, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from fast_ctc_decode_amd import dist as fdist
    from kat_cases import reference_style_rows

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    try:
        x = reference_style_rows(np.random.default_rng(3), n_reads * 120, 5).reshape(n_reads, 120, 5)
        bounds = fdist.shard_bounds(n_reads, world)
        counts = [hi - lo for lo, hi in bounds]
        lo, hi = bounds[rank]
        scratch = {}
        for _ in range(2):  # second call reuses the receive buffers
            res = fdist.decode_sharded(x[lo:hi], _oracle_decode, counts, dst=0, scratch=scratch)
        if rank == 0:
            r = res.cpu()
            np.savez(out_path, labels=r.labels, path=r.path, out_len=r.out_len, status=r.status)
        else:
            assert res is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_reads", [7, 8])
def test_sharded_decode_one_gather(tmp_path, n_reads):
    import torch.multiprocessing as mp
    from kat_cases import reference_style_rows

    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, _free_port(), n_reads, out), nprocs=2, join=True)
    got = np.load(out)
    x = reference_style_rows(np.random.default_rng(3), n_reads * 120, 5).reshape(n_reads, 120, 5)
    want = _oracle_decode(x)
    np.testing.assert_array_equal(got["out_len"], want.out_len.astype(np.int32))
    np.testing.assert_array_equal(got["status"], want.status)
    for i in range(n_reads):
        n = int(want.out_len[i])
        np.testing.assert_array_equal(got["labels"][i, :n], want.labels[i, :n])
        np.testing.assert_array_equal(got["path"][i, :n], want.path[i, :n].astype(np.int32))


@pytest.mark.parametrize("n_reads", [5, 11, 67])
def test_sharded_decode_world8_uneven_shards(tmp_path, n_reads):
    """BASELINE config 3's world size on CPU (gloo): 8 ranks, shards of different sizes -- empty ones included
    (5 reads over 8 ranks) -- one size all_reduce, ONE gather, rows in global read order on rank 0."""
    import torch.multiprocessing as mp
    from kat_cases import reference_style_rows

    out = str(tmp_path / "gathered8.npz")
    mp.spawn(_worker, args=(8, _free_port(), n_reads, out), nprocs=8, join=True)
    got = np.load(out)
    x = reference_style_rows(np.random.default_rng(3), n_reads * 120, 5).reshape(n_reads, 120, 5)
    want = _oracle_decode(x)
    np.testing.assert_array_equal(got["out_len"], want.out_len.astype(np.int32))
    np.testing.assert_array_equal(got["status"], want.status)
    for i in range(n_reads):
        n = int(want.out_len[i])
        np.testing.assert_array_equal(got["labels"][i, :n], want.labels[i, :n])
        np.testing.assert_array_equal(got["path"][i, :n], want.path[i, :n].astype(np.int32))


def test_shard_bounds():
    from fast_ctc_decode_amd.dist import shard_bounds
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_bounds(65536, 8)[-1] == (57344, 65536)
    b = shard_bounds(65537, 8)  # one read more than BASELINE config 3: the first shard takes it
    assert [hi - lo for lo, hi in b] == [8193] + [8192] * 7 and b[0][0] == 0 and b[-1][1] == 65537
    assert all(b[k][1] == b[k + 1][0] for k in range(7))
    assert [hi - lo for lo, hi in shard_bounds(0, 8)] == [0] * 8


def _random_result(B, W, seed, torch):
    from fast_ctc_decode_amd.api import BatchResult
    g = torch.Generator().manual_seed(seed)
    labels = torch.randint(1, 5, (B, W), dtype=torch.uint8, generator=g)
    path = torch.randint(0, W, (B, W), dtype=torch.int32, generator=g)
    out_len = torch.randint(0, W + 1, (B,), dtype=torch.int32, generator=g)
    status = torch.randint(0, 3, (B,), dtype=torch.int32, generator=g)
    out_len[status != 0] = 0
    return BatchResult(labels, path, out_len, status)


def _assert_same_used(a, b, W):
    import torch
    assert torch.equal(a.out_len, b.out_len) and torch.equal(a.status, b.status)
    mask = torch.arange(W)[None, :] < a.out_len[:, None]
    assert torch.equal(a.labels[mask], b.labels[mask]) and torch.equal(a.path[mask], b.path[mask])


def test_pack_unpack_roundtrip_wide_paths():
    """The gather payload stores path entries in two bytes below 65536 steps: values above 32767
    must survive, and reads of 65536+ steps fall back to four bytes."""
    import torch

    from fast_ctc_decode_amd import dist as fdist
    from fast_ctc_decode_amd.api import BatchResult

    for W in (50000, 70000):
        g = torch.Generator().manual_seed(W)
        B = 2
        labels = torch.randint(1, 5, (B, W), dtype=torch.uint8, generator=g)
        path = torch.randint(0, W, (B, W), dtype=torch.int32, generator=g)
        path[0, :3] = torch.tensor([0, 32768, W - 1], dtype=torch.int32)
        r = BatchResult(labels, path, torch.tensor([W, 3], dtype=torch.int32), torch.zeros(B, dtype=torch.int32))
        offs, total = fdist.result_total(r)
        assert total == W + 3
        nbytes = fdist.packed_nbytes(3, total, W)
        buf = fdist.pack_result(r, offs, nbytes)
        assert buf.numel() == nbytes
        back = fdist.unpack_results([buf], [B], W)
        _assert_same_used(r, back, W)


@pytest.mark.parametrize("B,W", [(3, 5), (1, 1), (7, 33), (4, 2), (5, 130)])
def test_pack_unpack_odd_shapes(B, W):
    """section offsets that are not 2- / 4-byte aligned (odd B, odd W) must round-trip."""
    import torch

    from fast_ctc_decode_amd import dist as fdist

    for seed in range(3):
        r = _random_result(B, W, 100 * B + W + seed, torch)
        offs, total = fdist.result_total(r)
        buf = fdist.pack_result(r, offs, fdist.packed_nbytes(B + 2, total, W) + 16 * seed)
        back = fdist.unpack_results([buf], [B], W)
        _assert_same_used(r, back, W)


def test_device_pack_kernels_match_the_host_layout():
    """csrc/pack.hip (through the lockstep emulation, tests/hipemu) writes byte for byte the buffer the
    numpy packer writes, and its unpack kernel inverts it -- the -m gpu twin runs on the MI355X."""
    import ctypes as C

    import torch

    from emu_util import emulated_kernels
    from fast_ctc_decode_amd import _native as nat
    from fast_ctc_decode_amd import dist as fdist

    with emulated_kernels():
        h = nat.default_handle()
        for B, W in ((3, 5), (9, 70), (130, 40), (2100, 3)):
            r = _random_result(B, W, B + W, torch)
            offs_host, total = fdist.result_total(r)
            nbytes = fdist.packed_nbytes(B, total, W)
            want = fdist.pack_result(r, offs_host, nbytes).numpy()
            offs = np.zeros(B + 1, np.uint64)
            h.check(h.lib.fcd_result_offsets_dev(h.ptr, r.out_len.data_ptr(), B, W, offs.ctypes.data))
            assert np.array_equal(offs.astype(np.int64), offs_host.numpy())
            assert h.lib.fcd_packed_result_bytes(B, total, 2) == nbytes
            got = np.zeros(nbytes, np.uint8)
            res = nat.Result(r.labels.data_ptr(), r.path.data_ptr(), None, r.out_len.data_ptr(), r.status.data_ptr(), W)
            h.check(h.lib.fcd_pack_results_dev(h.ptr, C.byref(res), B, 2, offs.ctypes.data, got.ctypes.data))
            used = 16 + 8 * B + ((total + 3) & ~3) + 2 * total   # padding bytes are unspecified
            lab_end = 16 + 8 * B + total
            assert np.array_equal(got[:lab_end], want[:lab_end])
            assert np.array_equal(got[16 + 8 * B + ((total + 3) & ~3):used], want[16 + 8 * B + ((total + 3) & ~3):used])
            back = fdist._to_torch(type(r)(np.zeros((B, W), np.uint8), np.zeros((B, W), np.int32),
                                           np.zeros(B, np.int32), np.zeros(B, np.int32)))
            res2 = nat.Result(back.labels.data_ptr(), back.path.data_ptr(), None, back.out_len.data_ptr(),
                              back.status.data_ptr(), W)
            work = np.zeros(B + 1, np.uint64)
            h.check(h.lib.fcd_unpack_results_dev(h.ptr, got.ctypes.data, B, work.ctypes.data, C.byref(res2)))
            _assert_same_used(r, back, W)


def test_unpack_rejects_a_shard_that_contradicts_the_counts():
    """a gathered buffer whose header names another read count, path width or more labels than the
    buffer holds must be an error, never an out-of-bounds read."""
    import torch

    from fast_ctc_decode_amd import dist as fdist

    B, W = 6, 20
    r = _random_result(B, W, 5, torch)
    offs, total = fdist.result_total(r)
    buf = fdist.pack_result(r, offs, fdist.packed_nbytes(B, total, W))
    for field, value in ((8, B + 1), (12, 3), (0, 10 ** 6)):
        bad = buf.clone()
        bad.numpy()[field:field + 4] = np.frombuffer(np.uint32(value).tobytes(), np.uint8)
        with pytest.raises(ValueError, match="gathered shard 0"):
            fdist.unpack_results([bad], [B], W)
    with pytest.raises(ValueError, match="gathered shard 0"):
        fdist.unpack_results([buf], [B - 1], W)


def test_gathered_unpack_and_c_abi_gather_under_the_emulator():
    """csrc/pack.hip's one-launch unpack of ALL shards (fcd_unpack_gathered_dev) and the C-ABI gather
    (csrc/comm.hip, fcd_gather_results_dev) with a communicator-less world of one, on the lockstep emulation:
    several shards of different sizes incl. empty ones, a corrupt header raises the flag and leaves its rows
    empty.  The RCCL leg of the same entry point runs in tests/test_gpu_dist.py."""
    import ctypes as C

    import torch

    from emu_util import emulated_kernels
    from fast_ctc_decode_amd import _native as nat
    from fast_ctc_decode_amd import dist as fdist
    from fast_ctc_decode_amd.api import BatchResult

    with emulated_kernels():
        h = nat.default_handle()
        W = 37
        counts = [5, 0, 130, 1, 0, 64]
        shards = [_random_result(B, W, 40 + k, torch) for k, B in enumerate(counts)]
        totals = [fdist.result_total(r) for r in shards]
        nbytes = max(fdist.packed_nbytes(max(counts), t, W) for _, t in totals)
        full = torch.zeros(nbytes * len(counts), dtype=torch.uint8)
        for k, (r, (offs, _)) in enumerate(zip(shards, totals)):
            if counts[k]:
                full[k * nbytes:(k + 1) * nbytes] = fdist.pack_result(r, offs, nbytes)
        n_total = sum(counts)
        first = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int64)

        def unpack(buf):
            out = BatchResult(torch.zeros((n_total, W), dtype=torch.uint8), torch.zeros((n_total, W), dtype=torch.int32),
                              torch.zeros(n_total, dtype=torch.int32), torch.zeros(n_total, dtype=torch.int32))
            work = torch.zeros(n_total + len(counts) + 1, dtype=torch.int64)
            bad = torch.zeros(1, dtype=torch.int32)
            res = nat.Result(out.labels.data_ptr(), out.path.data_ptr(), None, out.out_len.data_ptr(),
                             out.status.data_ptr(), W)
            h.check(h.lib.fcd_unpack_gathered_dev(h.ptr, buf.data_ptr(), nbytes, len(counts), first.data_ptr(), n_total,
                                                  work.data_ptr(), C.byref(res), bad.data_ptr()))
            return out, int(bad[0])

        out, bad = unpack(full)
        assert bad == 0
        row = 0
        for r, B in zip(shards, counts):
            view = BatchResult(out.labels[row:row + B], out.path[row:row + B], out.out_len[row:row + B],
                               out.status[row:row + B])
            _assert_same_used(r, view, W)
            row += B
        broken = full.clone()
        broken.numpy()[2 * nbytes + 8:2 * nbytes + 12] = np.frombuffer(np.uint32(129).tobytes(), np.uint8)
        out, bad = unpack(broken)
        assert bad == 3 and int(out.out_len[5:135].sum()) == 0  # shard 2 flagged, its rows empty
        _assert_same_used(shards[0], BatchResult(out.labels[:5], out.path[:5], out.out_len[:5], out.status[:5]), W)

        # the C-ABI gather, world of one, no communicator
        comm = C.c_void_p()
        h.check(h.lib.fcd_comm_wrap(h.ptr, None, 1, 0, C.byref(comm)))
        r = shards[2]
        B = counts[2]
        cnt = np.array([B], np.int64)
        for _ in range(2):  # the second call reuses the communicator's buffers
            got = BatchResult(torch.zeros((B, W), dtype=torch.uint8), torch.zeros((B, W), dtype=torch.int32),
                              torch.zeros(B, dtype=torch.int32), torch.zeros(B, dtype=torch.int32))
            src = nat.Result(r.labels.data_ptr(), r.path.data_ptr(), None, r.out_len.data_ptr(), r.status.data_ptr(), W)
            dst = nat.Result(got.labels.data_ptr(), got.path.data_ptr(), None, got.out_len.data_ptr(),
                             got.status.data_ptr(), W)
            h.check(h.lib.fcd_gather_results_dev(comm, C.byref(src), B, cnt.ctypes.data, 0, C.byref(dst)))
            h.check(h.lib.fcd_comm_synchronize(comm))
            _assert_same_used(r, got, W)
        assert h.lib.fcd_gather_results_dev(comm, C.byref(src), B - 1, cnt.ctypes.data, 0, C.byref(dst)) == nat.E_INVALID
        h.check(h.lib.fcd_comm_destroy(comm))
