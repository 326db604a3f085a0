"""developer scratch: the headline shape with `lengths` given (all equal to T: the ragged instantiations of the kernels, same
work as bench.py's), kernel time under both tie orders (python tools/dev/time_ragged.py [LIB])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from fast_ctc_decode_amd import _native as nat
if len(sys.argv) > 1:
    nat.LIB_PATH = os.path.abspath(sys.argv[1])
import fast_ctc_decode_amd as fcd
import bench
x = torch.from_numpy(bench.make_batch(1, 4096)).cuda()
lengths = np.full(4096, 4000, np.int64)
lengths[::7] = 3990  # (really ragged: a few reads end early)
out = []
for order in ("stable", "pdq178", "stable", "pdq178"):
    fcd.set_tie_order(order)
    r = fcd.beam_search_batch_raw(x, 5, 0.1, True, lengths=lengths)
    torch.cuda.synchronize()
    h = r._handle
    h.timing_reset()
    for _ in range(10):
        r = fcd.beam_search_batch_raw(x, 5, 0.1, True, lengths=lengths)
    torch.cuda.synchronize()
    out.append("%s %.3f" % (order, h.timing_mean_ms()[0]))
print(os.path.basename(nat.LIB_PATH), "ragged", " | ".join(out), flush=True)
