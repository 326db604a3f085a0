"""What the default order of equal probabilities (FCD_TIE_PDQ178) costs by beam width, reference-style rows: a launch alone
(HIP events around it) and sustained, with successive batches overlapping on the handle's internal streams
(fcd_set_overlap 4; wall clock over REPS batches / REPS).  python tools/tie_order_by_beam.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fast_ctc_decode_amd as fcd
from fast_ctc_decode_amd import _native as nat
import bench

REPS = int(os.environ.get("REPS", "12"))
OVERLAP = int(os.environ.get("OVERLAP", "4"))
for beam, batch in ((5, 4096), (8, 4096), (12, 4096), (32, 8192)):
    x = torch.from_numpy(bench.make_batch(1 if beam < 32 else 2, batch)).cuda()
    h = nat.default_handle()
    row = {}
    for order in ("stable", "pdq178"):
        fcd.set_tie_order(order)
        h.set_overlap(0)
        for _ in range(2):
            fcd.beam_search_batch_raw(x, beam, 0.1, True)
        torch.cuda.synchronize()
        h.timing_reset()
        for _ in range(4):
            fcd.beam_search_batch_raw(x, beam, 0.1, True)
        torch.cuda.synchronize()
        alone = h.timing_mean_ms()[0]
        h.set_overlap(OVERLAP)
        keep = [fcd.beam_search_batch_raw(x, beam, 0.1, True) for _ in range(REPS)]  # (warm: streams, regions, allocator)
        h.overlap_join()
        torch.cuda.synchronize()
        del keep
        t0 = time.perf_counter()
        keep = [fcd.beam_search_batch_raw(x, beam, 0.1, True) for _ in range(REPS)]
        h.overlap_join()
        torch.cuda.synchronize()
        sustained = (time.perf_counter() - t0) / REPS * 1e3
        del keep
        h.set_overlap(0)
        row[order] = (alone, sustained)
    a, b = row["stable"], row["pdq178"]
    print("beam %2d, %d reads: a launch alone  stable %.3f ms  pdq178 %.3f ms  (x%.2f) | sustained, overlap %d  stable %.3f ms  "
          "pdq178 %.3f ms  (x%.2f) = %.0f / %.0f reads/s" % (beam, batch, a[0], b[0], b[0] / a[0], OVERLAP, a[1], b[1], b[1] / a[1],
                                                           batch / a[1] * 1e3, batch / b[1] * 1e3), flush=True)
fcd.set_tie_order("pdq178")
