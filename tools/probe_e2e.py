"""End-to-end cost of the host batch API around the kernel (host numpy in -> Python objects out):

    python tools/probe_e2e.py [reads] [--grid]

Prints the PCIe legs on their own (pageable / page-locked upload of the batch), the device-resident search, the
fixed-stride fcd_beam_search_host call with and without the chunk pipeline, and the compiled module's
beam_search_batch with array / list / no paths.  --grid sweeps lanes x chunk sizes of the pipeline."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import fast_ctc_decode_amd as fcd
from fast_ctc_decode_amd import _native as nat
from fast_ctc_decode_amd import api


def best(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(args[0]) if args else 4096
    grid = "--grid" in sys.argv
    rng = np.random.default_rng(1)
    x = rng.random((B * 4000, 5), dtype=np.float32)
    x /= np.linalg.norm(x, ord=2, axis=1, keepdims=True)
    x = x.reshape(B, 4000, 5)
    mb = x.nbytes / 1e6
    xd = torch.empty(x.shape, dtype=torch.float32, device="cuda")
    xt = torch.from_numpy(x)
    ms_up, _ = best(lambda: xd.copy_(xt))
    xp = xt.pin_memory()
    ms_up_pin, _ = best(lambda: xd.copy_(xp, non_blocking=True))
    print("B=%d (%.0f MB): upload pageable %.2f ms (%.1f GB/s) | page-locked %.2f ms (%.1f GB/s)"
          % (B, mb, ms_up, mb / ms_up, ms_up_pin, mb / ms_up_pin))
    ms_dev, r = best(lambda: fcd.beam_search_batch_raw(xd, 5, 0.1))
    ms_cpu, rc = best(lambda: r.cpu())
    print("device-resident search %.2f ms | fixed-stride results to host (torch) %.2f ms" % (ms_dev, ms_cpu))
    want = rc.sequences("NACGT", paths="list")

    h = nat.default_handle()
    cm = api._compiled()

    def host_fixed():
        return fcd.beam_search_batch_raw(x, 5, 0.1)

    def report(tag):
        ms_fixed, _ = best(host_fixed, 2)
        ms_none, _ = best(lambda: cm.beam_search_batch(x, "NACGT", 5, 0.1, paths=None), 2)
        ms_arr, _ = best(lambda: cm.beam_search_batch(x, "NACGT", 5, 0.1, paths="array"), 3)
        ms_list, res = best(lambda: cm.beam_search_batch(x, "NACGT", 5, 0.1, paths="list"), 3)
        ok = all(a == b for a, b in zip(res, want))
        print("%-24s fcd_beam_search_host (fixed-stride arrays) %6.2f ms | beam_search_batch: no paths %6.2f ms, "
              "array paths %6.2f ms = %4.0fk reads/s, list paths %6.2f ms = %4.0fk reads/s | identical %s"
              % (tag, ms_fixed, ms_none, ms_arr, B / ms_arr, ms_list, B / ms_list, ok), flush=True)

    configs = [("one-shot (lanes=1)", 1, 0), ("default pipeline", 0, 0)]
    if grid:
        configs += [("lanes=%d chunk=%d" % (l, c), l, c) for l, c in
                    ((2, 2048), (2, 1024), (4, 512), (4, 2048), (8, 512), (8, 256), (3, 1408), (6, 704))]
    for tag, lanes, chunk in configs:
        h.set_host_pipeline(lanes, chunk, -1)
        cm._set_host_pipeline(lanes, chunk, -1)
        report(tag)
    h.set_host_pipeline(0, 0, -1)
    cm._set_host_pipeline(0, 0, -1)
    # a ragged batch as a Python LIST of per-read arrays (rows uniform in 2000..4000): the list goes through
    # fcd_beam_search_host_ptrs_begin -- chunks gathered by the lanes, no padded copy -- against padding it by hand
    rows = rng.integers(2000, 4001, B)
    reads = [np.ascontiguousarray(x[i, :rows[i]]) for i in range(B)]
    ms_rag, res_r = best(lambda: cm.beam_search_batch(reads, "NACGT", 5, 0.1, paths="array"), 3)

    def padded_by_hand():
        pad = np.zeros((B, 4000, 5), np.float32)
        for i, r in enumerate(reads):
            pad[i, :r.shape[0]] = r
        return cm.beam_search_batch(pad, "NACGT", 5, 0.1, lengths=rows, paths="array")
    ms_pad, res_p = best(padded_by_hand, 2)
    same = all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(res_r, res_p))
    print("ragged list of %d reads (rows 2000..4000, %.0f MB): beam_search_batch(list, array paths) %.2f ms = %.0fk reads/s | "
          "padded by hand + lengths %.2f ms = %.0fk reads/s | identical %s"
          % (B, sum(r.nbytes for r in reads) / 1e6, ms_rag, B / ms_rag, ms_pad, B / ms_pad, same), flush=True)
    # the per-read surface for scale: one call per read, one thread
    t0 = time.perf_counter()
    for i in range(64):
        fcd.beam_search(x[i], "NACGT", 5, 0.1)
    print("per-read beam_search calls, one thread: %.0f reads/s" % (64 / (time.perf_counter() - t0)))


if __name__ == "__main__":
    main()
