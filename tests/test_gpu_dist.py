"""The multi-GPU path on the box's one GPU: the RCCL ("nccl") backend is initialised with world_size 1 and
the product's shard -> decode -> pack -> size all_reduce -> ONE gather -> unpack code runs on device
tensors, so the communicator and the HIP pack / unpack kernels are exercised on fresh hardware every
round even without an 8-GPU node (tests/test_dist_gloo.py covers world_size 2 on CPU)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def nccl_world1():
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    if dist.is_initialized():
        yield dist
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=dev)
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_rccl_world1_sharded_decode(nccl_world1):
    import torch

    import fast_ctc_decode_amd as fcd
    from fast_ctc_decode_amd import dist as fdist
    from kat_cases import reference_style_rows

    B, T = 37, 300
    x = reference_style_rows(np.random.default_rng(11), B * T, 5).reshape(B, T, 5)
    x[5, 40] = np.nan   # one read fails: its status must travel, its row must stay empty
    xd = torch.from_numpy(x).cuda()
    local = fcd.beam_search_batch_raw(xd, 5, 0.1, True).cpu()
    scratch = {}
    for _ in range(2):   # the second call reuses the buffers
        got = fdist.beam_search_sharded(xd, [B], 5, 0.1, True, dst=0, scratch=scratch)
    torch.cuda.synchronize()
    got = got.cpu()
    np.testing.assert_array_equal(got.out_len, local.out_len)
    np.testing.assert_array_equal(got.status, local.status)
    assert int(local.status[5]) == 2 and int(got.out_len[5]) == 0
    for i in range(B):
        n = int(local.out_len[i])
        np.testing.assert_array_equal(got.labels[i, :n], local.labels[i, :n])
        np.testing.assert_array_equal(got.path[i, :n], local.path[i, :n])


@pytest.mark.parametrize("B,W", [(3, 5), (130, 40), (4096, 257), (2100, 3)])
def test_hip_pack_kernels_match_the_host_layout(B, W):
    """csrc/pack.hip on the GPU writes the bytes the numpy packer writes; unpack inverts it."""
    torch = pytest.importorskip("torch")

    from fast_ctc_decode_amd import dist as fdist
    from fast_ctc_decode_amd.api import BatchResult
    from test_dist_gloo import _assert_same_used, _random_result

    r = _random_result(B, W, B + W, torch)
    offs_host, total = fdist.result_total(r)
    nbytes = fdist.packed_nbytes(B, total, W)
    want = fdist.pack_result(r, offs_host, nbytes).numpy()
    rd = BatchResult(r.labels.cuda(), r.path.cuda(), r.out_len.cuda(), r.status.cuda())
    offs, total_d = fdist.result_total(rd)
    assert total_d == total and torch.equal(offs.cpu(), offs_host)
    got_d = fdist.pack_result(rd, offs, nbytes)
    got = got_d.cpu().numpy()
    lab0 = 16 + 8 * B
    p0 = lab0 + ((total + 3) & ~3)
    assert np.array_equal(got[:lab0 + total], want[:lab0 + total])
    assert np.array_equal(got[p0:p0 + 2 * total], want[p0:p0 + 2 * total])
    back = fdist.unpack_results([got_d], [B], W)
    torch.cuda.synchronize()
    _assert_same_used(r, BatchResult(back.labels.cpu(), back.path.cpu(), back.out_len.cpu(), back.status.cpu()), W)


def test_c_abi_gather_over_rccl_world1():
    """csrc/comm.hip: fcd_comm_unique_id / fcd_comm_create (ncclCommInitRank through the dlopen'ed RCCL),
    fcd_gather_results_dev (offsets + pack, ncclAllReduce(MAX) of the size, ONE ncclGather, one-launch unpack) and
    fcd_comm_synchronize on this box's GPU with a world of one: the entry point a non-Python host binds
     runs through RCCL in every driver round."""
    import ctypes as C

    torch = pytest.importorskip("torch")

    import fast_ctc_decode_amd as fcd
    from fast_ctc_decode_amd import _native as nat
    from kat_cases import reference_style_rows

    B, T = 300, 500
    x = reference_style_rows(np.random.default_rng(21), B * T, 5).reshape(B, T, 5)
    x[7, 100] = np.nan
    xd = torch.from_numpy(x).cuda()
    r = fcd.beam_search_batch_raw(xd, 5, 0.1, True)
    torch.cuda.synchronize()
    h = nat.Handle(0)
    try:
        ident = (C.c_uint8 * 128)()
        h.check(h.lib.fcd_comm_unique_id(ident))
        comm = C.c_void_p()
        h.check(h.lib.fcd_comm_create(h.ptr, 1, 0, ident, C.byref(comm)))
        counts = np.array([B], np.int64)
        for _ in range(3):  # later calls reuse the communicator's buffers
            labels = torch.full((B, T), 255, dtype=torch.uint8, device="cuda")
            path = torch.full((B, T), -1, dtype=torch.int32, device="cuda")
            out_len = torch.full((B,), -1, dtype=torch.int32, device="cuda")
            status = torch.full((B,), -1, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            src = nat.Result(r.labels.data_ptr(), r.path.data_ptr(), None, r.out_len.data_ptr(), r.status.data_ptr(), T)
            dst = nat.Result(labels.data_ptr(), path.data_ptr(), None, out_len.data_ptr(), status.data_ptr(), T)
            h.check(h.lib.fcd_gather_results_dev(comm, C.byref(src), B, counts.ctypes.data, 0, C.byref(dst)))
            h.check(h.lib.fcd_comm_synchronize(comm))
            assert torch.equal(out_len, r.out_len) and torch.equal(status, r.status)
            mask = torch.arange(T, device="cuda")[None, :] < out_len[:, None]
            assert torch.equal(labels[mask], r.labels[mask]) and torch.equal(path[mask], r.path[mask])
            assert int(status[7]) == 2 and int(out_len[7]) == 0
        h.check(h.lib.fcd_comm_destroy(comm))
    finally:
        h.close()


def test_one_launch_unpack_of_many_shards():
    """fcd_unpack_gathered_dev on the GPU: eight shards of different sizes (an empty one among them) packed
    back to back, as ncclGather delivers them, come out in global read order; fast_ctc_decode_amd.dist reuses
    the result arrays across calls."""
    torch = pytest.importorskip("torch")

    from fast_ctc_decode_amd import dist as fdist
    from fast_ctc_decode_amd.api import BatchResult
    from test_dist_gloo import _assert_same_used, _random_result

    W = 4000
    counts = [700, 1, 0, 512, 33, 1024, 2, 64]
    shards = [_random_result(B, W, 70 + k, torch) for k, B in enumerate(counts)]
    dev = [BatchResult(r.labels.cuda(), r.path.cuda(), r.out_len.cuda(), r.status.cuda()) for r in shards]
    totals = [fdist.result_total(r) for r in dev]
    nbytes = max(fdist.packed_nbytes(max(counts), t, W) for _, t in totals)
    full = torch.zeros(nbytes * len(counts), dtype=torch.uint8, device="cuda")
    bufs = [full[k * nbytes:(k + 1) * nbytes] for k in range(len(counts))]
    for k, (r, (offs, _)) in enumerate(zip(dev, totals)):
        if counts[k]:
            fdist.pack_result(r, offs, nbytes, out=bufs[k])
    scratch = {}
    first_out = None
    for _ in range(2):
        out = fdist.unpack_results(bufs, counts, W, scratch=scratch)
        fdist.check_gather(scratch)
        first_out = first_out or out
        assert out.labels.data_ptr() == first_out.labels.data_ptr()  # no new world x B x W allocation per step
        oc = out.cpu()
        oc = BatchResult(torch.from_numpy(oc.labels), torch.from_numpy(oc.path), torch.from_numpy(oc.out_len),
                         torch.from_numpy(oc.status))
        row = 0
        for r, B in zip(shards, counts):
            _assert_same_used(r, BatchResult(oc.labels[row:row + B], oc.path[row:row + B], oc.out_len[row:row + B],
                                             oc.status[row:row + B]), W)
            row += B
    # a shard whose header names another read count is reported, not read out of bounds
    full[nbytes * 3 + 8:nbytes * 3 + 12] = torch.tensor([1, 2, 0, 0], dtype=torch.uint8, device="cuda")
    fdist.unpack_results(bufs, counts, W, scratch=scratch)
    with pytest.raises(ValueError, match="gathered shard 3"):
        fdist.check_gather(scratch)
