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


def test_shard_bounds():
    from fast_ctc_decode_amd.dist import shard_bounds
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_bounds(65536, 8)[-1] == (57344, 65536)


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
        buf = fdist.pack_result(r, pad_reads=3)
        assert buf.numel() == fdist.packed_nbytes(3, W)
        back = fdist.unpack_results([buf], [B], W, 3)
        assert torch.equal(back.labels, labels) and torch.equal(back.path, path)
        assert torch.equal(back.out_len, r.out_len) and torch.equal(back.status, r.status)
