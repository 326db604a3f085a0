"""Multi-GPU decoding: reads shard embarrassingly, one process per GPU, ONE gather of results.

Reads are independent units (the reference decodes one read per call, src/lib.rs:318-365), so a
batch is partitioned into contiguous per-rank shards and every rank decodes its shard with no
collective inside the search.  The decoded (labels, path, out_len, status) of a shard are packed
into one contiguous byte buffer and moved to the destination rank with a single
`torch.distributed.gather` -- RCCL over xGMI with the "nccl" backend on MI355X nodes, gloo on CPU
(tests/test_dist_gloo.py runs this exact code path with world_size 2).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a gather to rank 0 uses each peer's direct
link once, so it is per-link bound; at BASELINE config 2 the payload is ~49 MB per rank (labels u8 +
path u16 at fixed stride T = 4000).
"""
import numpy as np

from .api import BatchResult


def shard_bounds(n_reads, world):
    """Contiguous, balanced [lo, hi) ranges: the first n_reads % world shards get one extra read."""
    q, r = divmod(int(n_reads), int(world))
    out, lo = [], 0
    for i in range(world):
        hi = lo + q + (1 if i < r else 0)
        out.append((lo, hi))
        lo = hi
    return out


def _path_bytes(width):
    # path entries are time indices < width: two bytes are enough for reads shorter than 65536 steps
    return 2 if width <= 65535 else 4


def packed_nbytes(n_reads, width):
    # labels u8 [B,W] | path u16/u32 [B,W] | out_len u32 [B] | status i32 [B]
    return n_reads * width + _path_bytes(width) * n_reads * width + 4 * n_reads + 4 * n_reads


def pack_result(r, pad_reads, out=None):
    """BatchResult (torch tensors, any device) -> one uint8 tensor of packed_nbytes(pad_reads, W).
    `out` (optional) is a reusable buffer of that size."""
    import torch

    labels = r.labels
    B, W = labels.shape
    dev = labels.device
    nbytes = packed_nbytes(pad_reads, W)
    if out is not None and out.numel() == nbytes and out.device == dev:
        buf = out
    else:
        buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    o = 0
    buf[o:o + B * W] = labels.reshape(-1)
    o = pad_reads * W
    pb = _path_bytes(W)
    path = r.path.contiguous() if pb == 4 else r.path.to(torch.int16)  # low 16 bits, exact below 65536
    buf[o:o + pb * B * W] = path.view(torch.uint8).reshape(-1)
    o += pb * pad_reads * W
    buf[o:o + 4 * B] = r.out_len.contiguous().view(torch.uint8).reshape(-1)
    o += 4 * pad_reads
    buf[o:o + 4 * B] = r.status.contiguous().view(torch.uint8).reshape(-1)
    return buf


def unpack_results(bufs, counts, width, pad_reads):
    """Inverse of pack_result for the gathered per-rank buffers -> one BatchResult (same device)."""
    import torch

    labels, path, out_len, status = [], [], [], []
    W = width
    for buf, B in zip(bufs, counts):
        o = 0
        labels.append(buf[o:o + B * W].reshape(B, W))
        o = pad_reads * W
        pb = _path_bytes(W)
        if pb == 4:
            path.append(buf[o:o + 4 * B * W].view(torch.int32).reshape(B, W))
        else:
            path.append(buf[o:o + 2 * B * W].view(torch.int16).reshape(B, W).to(torch.int32) & 0xFFFF)
        o += pb * pad_reads * W
        out_len.append(buf[o:o + 4 * B].view(torch.int32))
        o += 4 * pad_reads
        status.append(buf[o:o + 4 * B].view(torch.int32))
    return BatchResult(torch.cat(labels), torch.cat(path), torch.cat(out_len), torch.cat(status))


def _to_torch(r):
    import torch

    def t(a, dtype):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a).view(dtype))
        return a
    return BatchResult(t(r.labels, np.uint8), t(r.path, np.int32), t(r.out_len, np.int32),
                       t(r.status, np.int32))


def gather_batch_result(r, counts, dst=0, group=None, scratch=None):
    """ONE collective: gather every rank's packed shard result on `dst`.

    r       this rank's BatchResult (numpy or torch)
    counts  reads per rank (len == world size), e.g. from shard_bounds
    scratch optional dict reused across calls to avoid re-allocating the receive buffers
    Returns the concatenated BatchResult on dst, None elsewhere."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    r = _to_torch(r)
    pad = max(counts)
    W = r.labels.shape[1]
    send = pack_result(r, pad, out=None if scratch is None else scratch.get("send"))
    if scratch is not None:
        scratch["send"] = send
    recv = None
    if rank == dst:
        key = (pad, W, send.device)
        if scratch is not None and scratch.get("key") == key:
            recv = scratch["bufs"]
        else:
            recv = [torch.empty_like(send) for _ in range(world)]
            if scratch is not None:
                scratch["key"], scratch["bufs"] = key, recv
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return unpack_results(recv, counts, W, pad)


def decode_sharded(x_local, decode_fn, counts, dst=0, group=None, scratch=None):
    """Decode this rank's shard with `decode_fn(x_local) -> BatchResult`, then one gather to dst."""
    r = decode_fn(x_local)
    return gather_batch_result(r, counts, dst=dst, group=group, scratch=scratch)


def beam_search_sharded(x_local, counts, beam_size=5, beam_cut_threshold=0.0, collapse_repeats=True,
                        dst=0, group=None, scratch=None):
    """Every rank passes its own shard (a torch tensor on its GPU, or numpy); rank `dst` gets the
    whole batch's BatchResult in global read order."""
    from . import api

    return decode_sharded(
        x_local,
        lambda x: api.beam_search_batch_raw(x, beam_size, beam_cut_threshold, collapse_repeats),
        counts, dst=dst, group=group, scratch=scratch)
