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
