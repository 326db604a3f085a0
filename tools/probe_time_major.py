"""Basecaller networks emit time-major scores, (T, B, N) -- often float16.  The batch API takes any strides, so such a
tensor is passed as a permuted VIEW (no copy): what does that layout cost against read-major (B, T, N)?

    python tools/probe_time_major.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import fast_ctc_decode_amd as fcd


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    r = None
    h = None
    for _ in range(2):
        r = fn()
    torch.cuda.synchronize()
    h = r._handle
    h.timing_reset()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    ms, _ = h.timing_mean_ms()
    return ms, r


def main():
    B, T, N = 4096, 4000, 5
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    x = torch.rand((B, T, N), generator=g, device="cuda", dtype=torch.float32)
    x /= torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)
    for dt in (torch.float32, torch.float16):
        xr = x.to(dt).contiguous()                       # read-major (B, T, N)
        xt = xr.permute(1, 0, 2).contiguous()            # time-major storage (T, B, N)
        view = xt.permute(1, 0, 2)                       # ... seen as (B, T, N): a strided view, no copy
        for name, fn_r, fn_t in (
                ("beam_search(5, 0.1)", lambda: fcd.beam_search_batch_raw(xr, 5, 0.1), lambda: fcd.beam_search_batch_raw(view, 5, 0.1)),
                ("viterbi_search", lambda: fcd.viterbi_search_batch_raw(xr), lambda: fcd.viterbi_search_batch_raw(view))):
            ms_r, rr = timed(fn_r)
            ms_t, rt = timed(fn_t)
            a, b = rr.cpu(), rt.cpu()
            same = bool((a.out_len == b.out_len).all()) and all(
                (a.labels[i, :int(a.out_len[i])] == b.labels[i, :int(a.out_len[i])]).all() and
                (a.path[i, :int(a.out_len[i])] == b.path[i, :int(a.out_len[i])]).all() for i in range(0, B, 37))
            print("%-8s %-20s read-major %.3f ms | time-major view %.3f ms (x%.2f) | identical %s"
                  % (str(dt).replace("torch.", ""), name, ms_r, ms_t, ms_t / ms_r, same), flush=True)


def crf():
    """CRF scores, (T, B, S, N) time-major storage seen as (B, T, S, N)"""
    B, T, S, N = 4096, 4000, 4, 5
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    x = torch.rand((B, T, S, N), generator=g, device="cuda", dtype=torch.float32)
    init = torch.rand((B, S), generator=g, device="cuda")
    xt = x.permute(1, 0, 2, 3).contiguous()
    view = xt.permute(1, 0, 2, 3)
    for name, fn_r, fn_t in (
            ("crf_beam_search(5, 0.0)", lambda: fcd.crf_beam_search_batch_raw(x, init, 5, 0.0), lambda: fcd.crf_beam_search_batch_raw(view, init, 5, 0.0)),
            ("crf_greedy_search", lambda: fcd.crf_greedy_search_batch_raw(x, init), lambda: fcd.crf_greedy_search_batch_raw(view, init))):
        ms_r, rr = timed(fn_r, 3)
        ms_t, rt = timed(fn_t, 3)
        a, b = rr.cpu(), rt.cpu()
        same = bool((a.out_len == b.out_len).all()) and all(
            (a.labels[i, :int(a.out_len[i])] == b.labels[i, :int(a.out_len[i])]).all() for i in range(0, B, 37))
        print("float32  %-23s read-major %.3f ms | time-major view %.3f ms (x%.2f) | identical %s"
              % (name, ms_r, ms_t, ms_t / ms_r, same), flush=True)


if __name__ == "__main__":
    main()
    crf()
