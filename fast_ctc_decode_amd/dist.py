"""Multi-GPU decoding: reads shard embarrassingly, one process per GPU, ONE gather of results.

Reads are independent units (the reference decodes one read per call, src/lib.rs:318-365), so a
batch is partitioned into contiguous per-rank shards and every rank decodes its shard with no
collective inside the search.  A shard's decoded (labels, path, out_len, status) are packed into one
contiguous byte buffer holding only the USED prefix of every fixed-stride row (csrc/pack.hip; ~48 % of a
row at BASELINE config 2) and moved to the destination rank with a single `torch.distributed.gather`
-- RCCL over xGMI with the "nccl" backend on MI355X nodes, gloo on CPU (tests/test_dist_gloo.py runs
this code path with world_size 2).  A gather needs equally sized buffers, so the ranks first agree on
the largest payload with an 8-byte all_reduce(MAX); that is the only other collective.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a gather to rank 0 uses each peer's direct
link once, so it is per-link bound; at BASELINE config 2 the payload is ~24 MB per rank (4096 reads x
~1940 labels x (u8 label + u16 time)) instead of 49 MB for fixed-stride rows.
"""
import ctypes as C

import numpy as np

from . import _native as nat
from .api import BatchResult

_HEADER = 16


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


def packed_nbytes(n_reads, total_labels, width):
    """Bytes of the wire buffer (csrc/pack.hip): header | out_len u32[n] | status i32[n] | labels u8[total]
    (padded to 4) | path u16/u32[total], rounded up to 16."""
    pb = _path_bytes(width)
    b = _HEADER + 8 * n_reads + ((total_labels + 3) & ~3) + total_labels * pb
    return (b + 15) & ~15


def _is_device(t):
    return hasattr(t, "is_cuda") and t.is_cuda


def _handle_for(t):
    import torch

    h = nat.default_handle(t.device.index or 0)
    h.set_stream(torch.cuda.current_stream(t.device).cuda_stream)
    return h


def _result_struct(r, width):
    return nat.Result(r.labels.data_ptr(), r.path.data_ptr() if r.path is not None else None, None,
                      r.out_len.data_ptr(), r.status.data_ptr() if r.status is not None else None, width)


def result_total(r):
    """-> (offsets, total): prefix sums of out_len (torch int64, n+1 entries, same device) and their last
    element as a Python int (one small device -> host read)."""
    import torch

    B, W = r.labels.shape
    if _is_device(r.labels):
        h = _handle_for(r.labels)
        offs = torch.empty(B + 1, dtype=torch.int64, device=r.labels.device)
        try:
            h.check(h.lib.fcd_result_offsets_dev(h.ptr, r.out_len.data_ptr(), B, W, offs.data_ptr()))
        finally:
            h.reset_stream()  # the thread's default handle does not stay bound to the caller's (comm) stream
    else:
        lens = torch.clamp(r.out_len.to(torch.int64), max=W)
        offs = torch.zeros(B + 1, dtype=torch.int64)
        offs[1:] = torch.cumsum(lens, 0)
    return offs, int(offs[-1].item())


def pack_result(r, offsets, nbytes, out=None):
    """BatchResult (torch tensors, any device) -> one uint8 tensor of `nbytes` (>= packed_nbytes of this
    shard) in the wire format.  `out` (optional) is a reusable buffer."""
    import torch

    B, W = r.labels.shape
    dev = r.labels.device
    pb = _path_bytes(W)
    if out is not None and out.numel() == nbytes and out.device == dev:
        buf = out
    else:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if _is_device(r.labels):
        h = _handle_for(r.labels)
        res = _result_struct(r, W)
        try:
            h.check(h.lib.fcd_pack_results_dev(h.ptr, C.byref(res), B, pb, offsets.data_ptr(), buf.data_ptr()))
        finally:
            h.reset_stream()
        return buf
    # host tensors (gloo): the same layout with numpy
    o = offsets.numpy()
    total = int(o[-1])
    b = buf.numpy()
    b[:8] = np.frombuffer(np.uint64(total).tobytes(), np.uint8)
    b[8:12] = np.frombuffer(np.uint32(B).tobytes(), np.uint8)
    b[12:16] = np.frombuffer(np.uint32(pb).tobytes(), np.uint8)
    lens = (o[1:] - o[:-1]).astype(np.uint32)
    b[16:16 + 4 * B] = lens.view(np.uint8)
    b[16 + 4 * B:16 + 8 * B] = r.status.numpy().astype(np.int32).view(np.uint8)
    mask = np.arange(W)[None, :] < lens[:, None]
    lab0 = _HEADER + 8 * B
    b[lab0:lab0 + total] = r.labels.numpy()[mask]
    p0 = lab0 + ((total + 3) & ~3)
    pth = r.path.numpy()[mask]
    b[p0:p0 + total * pb] = pth.astype(np.uint16 if pb == 2 else np.uint32).view(np.uint8)
    return buf


def _check_header(b, B, width, capacity, rank):
    """A packed shard must describe exactly the reads the receiver expects: anything else would make the
    unpacking read or write out of bounds."""
    total = int(b[:8].view(np.uint64)[0])
    n_in = int(b[8:12].view(np.uint32)[0])
    pb = int(b[12:16].view(np.uint32)[0]) & 0xFF
    if n_in != B or pb not in (2, 4) or _HEADER + 8 * B + ((total + 3) & ~3) + total * pb > capacity \
            or total > B * width:
        raise ValueError("gathered shard %d: header (reads %d, path bytes %d, labels %d) contradicts the expected "
                         "%d reads of width %d in %d bytes" % (rank, n_in, pb, total, B, width, capacity))
    return total, pb


def check_gather(scratch):
    """Raises if the last device-side unpack met a shard whose header contradicted the read counts (the flag is
    written by the unpack kernels; gather_batch_result looks at it at its next call, this looks now)."""
    bad = None if scratch is None else scratch.get("bad")
    if bad is not None:
        v = int(bad.item())
        if v:
            bad.zero_()
            raise ValueError("gathered shard %d: header contradicts the expected read count / buffer size" % (v - 1))


def unpack_results(bufs, counts, width, scratch=None):
    """Inverse of pack_result for the gathered per-rank buffers -> one BatchResult (same device) with
    fixed-stride rows of `width` entries, in global read order.

    On a GPU the shards are unpacked by ONE pair of launches (fcd_unpack_gathered_dev) when `bufs` are
    consecutive slices of one allocation (as gather_batch_result makes them), into result arrays that `scratch`
    keeps across calls (rank 0 does not allocate world x B x width x 5 bytes per step)."""
    import torch

    n_total = int(sum(counts))
    dev = bufs[0].device
    world = len(bufs)
    on_gpu = _is_device(bufs[0])
    have = None if scratch is None else scratch.get("result")
    if have is not None and have.labels.shape == (n_total, width) and have.labels.device == dev:
        out = have
    else:
        # (rows are only defined up to out_len: on a GPU the 5 bytes x width x reads are not cleared first)
        make = torch.empty if on_gpu else torch.zeros
        out = BatchResult(make((n_total, width), dtype=torch.uint8, device=dev),
                          make((n_total, width), dtype=torch.int32, device=dev),
                          torch.zeros(n_total, dtype=torch.int32, device=dev),
                          torch.zeros(n_total, dtype=torch.int32, device=dev))
        if scratch is not None:
            scratch["result"] = out
    labels, path, out_len, status = out.labels, out.path, out.out_len, out.status
    if on_gpu:
        nbytes = bufs[0].numel()
        contiguous = all(b.numel() == nbytes and b.data_ptr() == bufs[0].data_ptr() + k * nbytes
                         for k, b in enumerate(bufs))
        h = _handle_for(bufs[0])
        try:
            key = tuple(int(c) for c in counts)
            if scratch is not None and scratch.get("first_key") == key and scratch["first"].device == dev:
                first = scratch["first"]
            else:
                first = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int64, device=dev)
                if scratch is not None:
                    scratch["first"], scratch["first_key"] = first, key
            offs = torch.empty(n_total + world + 1, dtype=torch.int64, device=dev)
            bad = scratch.get("bad") if scratch is not None else None
            if bad is None or bad.device != dev:
                bad = torch.zeros(1, dtype=torch.int32, device=dev)
                if scratch is not None:
                    scratch["bad"] = bad
            res = _result_struct(out, width)
            if contiguous:
                h.check(h.lib.fcd_unpack_gathered_dev(h.ptr, bufs[0].data_ptr(), nbytes, world, first.data_ptr(),
                                                      n_total, offs.data_ptr(), C.byref(res), bad.data_ptr()))
            else:  # separate allocations: shard by shard, each as a "world" of one
                row = 0
                for k, (buf, B) in enumerate(zip(bufs, counts)):
                    if B:
                        view = BatchResult(labels[row:row + B], path[row:row + B], out_len[row:row + B],
                                           status[row:row + B])
                        f1 = torch.tensor([0, B], dtype=torch.int64, device=dev)
                        o1 = torch.empty(B + 2, dtype=torch.int64, device=dev)
                        r1 = _result_struct(view, width)
                        h.check(h.lib.fcd_unpack_gathered_dev(h.ptr, buf.data_ptr(), buf.numel(), 1, f1.data_ptr(), B,
                                                              o1.data_ptr(), C.byref(r1), bad.data_ptr()))
                    row += B
            out._keep = (offs, first, bad)
            if scratch is None:
                v = int(bad.item())
                if v:
                    raise ValueError("gathered shard %d: header contradicts the expected read count / buffer size"
                                     % (v - 1))
        finally:
            h.reset_stream()
        return out
    row = 0
    for k, (buf, B) in enumerate(zip(bufs, counts)):
        if B == 0:
            continue
        b = buf.numpy()
        total, pb = _check_header(b, B, width, b.size, k)
        lens = b[16:16 + 4 * B].view(np.uint32).copy()
        if int(lens.sum()) != total or (lens > width).any():
            raise ValueError("gathered shard %d: lengths do not add up to the header's total" % k)
        out_len[row:row + B] = torch.from_numpy(lens.astype(np.int32))
        status[row:row + B] = torch.from_numpy(b[16 + 4 * B:16 + 8 * B].view(np.int32).copy())
        mask = np.arange(width)[None, :] < lens[:, None]
        lab0 = _HEADER + 8 * B
        lab = np.zeros((B, width), np.uint8)
        lab[mask] = b[lab0:lab0 + total]
        p0 = lab0 + ((total + 3) & ~3)
        pth = np.zeros((B, width), np.int32)
        pth[mask] = b[p0:p0 + total * pb].view(np.uint16 if pb == 2 else np.uint32).astype(np.int32)
        labels[row:row + B] = torch.from_numpy(lab)
        path[row:row + B] = torch.from_numpy(pth)
        row += B
    return out


def _to_torch(r):
    import torch

    def t(a, dtype):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a).view(dtype))
        return a.contiguous()
    return BatchResult(t(r.labels, np.uint8), t(r.path, np.int32), t(r.out_len, np.int32),
                       t(r.status, np.int32))


def gather_batch_result(r, counts, dst=0, group=None, scratch=None):
    """Gather every rank's packed shard result on `dst`: an 8-byte all_reduce(MAX) of the payload size,
    then ONE gather of the compact buffers.

    r       this rank's BatchResult (numpy or torch)
    counts  reads per rank (len == world size), e.g. from shard_bounds
    scratch optional dict reused across calls to avoid re-allocating the buffers.  WITH a scratch dict the BatchResult
            returned on dst LIVES IN IT: the next call with the same dict overwrites its labels / path / out_len /
            status arrays -- copy what must outlive the next step (or pass a fresh dict per result you keep) -- and
            a shard header that contradicts `counts` is reported by the NEXT call, or by check_gather(scratch):
            call check_gather(scratch) after the last step.
    Returns the concatenated BatchResult on dst, None elsewhere."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    check_gather(scratch)  # the previous call's header check (no extra wait: this call reads a size back anyway)
    r = _to_torch(r)
    W = r.labels.shape[1]
    offsets, total = result_total(r)
    size = torch.tensor([packed_nbytes(max(counts), total, W)], dtype=torch.int64, device=r.labels.device)
    if world > 1:
        dist.all_reduce(size, op=dist.ReduceOp.MAX, group=group)
    nbytes = int(size.item())
    # buffers are reused while the agreed size does not grow past what was allocated
    send = None
    if scratch is not None and scratch.get("send") is not None and scratch["send"].numel() >= nbytes \
            and scratch["send"].device == r.labels.device:
        send = scratch["send"][:nbytes]
    send = pack_result(r, offsets, nbytes, out=send)
    if scratch is not None and (scratch.get("send") is None or scratch["send"].numel() < nbytes):
        scratch["send"] = send
    recv = None
    if rank == dst:
        # ONE allocation holds all shards back to back, so that they can be unpacked by one pair of launches
        have = None if scratch is None else scratch.get("recv")
        if have is None or have.numel() < nbytes * world or have.device != send.device:
            have = torch.empty(nbytes * world + nbytes // 4, dtype=torch.uint8, device=send.device)
            if scratch is not None:
                scratch["recv"] = have
        recv = [have[k * nbytes:(k + 1) * nbytes] for k in range(world)]
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return unpack_results(recv, counts, W, scratch=scratch)


def decode_sharded(x_local, decode_fn, counts, dst=0, group=None, scratch=None):
    """Decode this rank's shard with `decode_fn(x_local) -> BatchResult`, then one gather to dst.  (With `scratch`
    the result on dst is valid until the next call with the same dict -- see gather_batch_result; call
    check_gather(scratch) after the last step.)"""
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
