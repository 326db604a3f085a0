"""developer scratch: config-5 kernel time (1024 pairs, T = 2000, band +-64; logsumexp and max) of a given build of the
library (python tools/dev/time_duplex.py [LIB])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from fast_ctc_decode_amd import _native as nat
if len(sys.argv) > 1:
    nat.LIB_PATH = os.path.abspath(sys.argv[1])
import fast_ctc_decode_amd as fcd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from duplex_account import gen
B, T, w = 1024, 2000, 64
x1, x2 = gen(B, T, 5, 4), gen(B, T, 5, 5)
i = np.arange(T)
env = np.stack([np.maximum(0, i - w), np.minimum(T, i + w)], 1).astype(np.uint64)
envs = torch.from_numpy(np.broadcast_to(env, (B, T, 2)).copy().view(np.int64)).cuda()
h = nat.default_handle()
out = []
for mode, name in ((0, "logsumexp"), (1, "max")):
    ms = []
    for _ in range(4):
        r = fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, logadd_mode=mode)
        torch.cuda.synchronize()
        ms.append(h.last_kernel_ms())
    out.append("%s %s" % (name, " ".join("%.2f" % m for m in ms[1:])))
print(os.path.basename(nat.LIB_PATH), " | ".join(out), flush=True)
