"""developer scratch: config-2 kernel time of a given build of the library (python tools/dev/time_variant.py [LIB])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fast_ctc_decode_amd import _native as nat
if len(sys.argv) > 1:
    nat.LIB_PATH = os.path.abspath(sys.argv[1])
import fast_ctc_decode_amd as fcd
import bench
x = torch.from_numpy(bench.make_batch(1, 4096)).cuda()
for order in ("stable", "pdq178", "stable", "pdq178"):
    fcd.set_tie_order(order)
    r = fcd.beam_search_batch_raw(x, 5, 0.1, True)
    torch.cuda.synchronize()
    h = r._handle
    h.timing_reset()
    for _ in range(10):
        r = fcd.beam_search_batch_raw(x, 5, 0.1, True)
    torch.cuda.synchronize()
    print(os.path.basename(nat.LIB_PATH), order, "%.3f ms" % h.timing_mean_ms()[0], flush=True)
