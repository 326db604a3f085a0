"""developer scratch: config-2 kernel time of a given build of the library, reference-style and peaky rows, both tie
orders (python tools/dev/time_variant.py [LIB])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fast_ctc_decode_amd import _native as nat
if len(sys.argv) > 1:
    nat.LIB_PATH = os.path.abspath(sys.argv[1])
import fast_ctc_decode_amd as fcd
import bench
BEAM = int(os.environ.get('BEAM', '5'))
BATCH = int(os.environ.get('BATCH', '4096'))
REPS = int(os.environ.get('REPS', '10'))
for name, gen in (("reference", bench.make_batch), ("peaky", bench.make_batch_peaky)):
    x = torch.from_numpy(gen(1, BATCH)).cuda()
    out = []
    for order in ("stable", "pdq178", "stable", "pdq178"):
        fcd.set_tie_order(order)
        r = fcd.beam_search_batch_raw(x, BEAM, 0.1, True)
        torch.cuda.synchronize()
        h = r._handle
        h.timing_reset()
        for _ in range(REPS):
            r = fcd.beam_search_batch_raw(x, BEAM, 0.1, True)
        torch.cuda.synchronize()
        out.append("%s %.3f" % (order, h.timing_mean_ms()[0]))
    print(os.path.basename(nat.LIB_PATH), name, " | ".join(out), flush=True)
