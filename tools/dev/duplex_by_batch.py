"""developer scratch: config-5-shaped pairs (T = 2000, band +-64, beam 5, thr 0.1), kernel time and pairs/s by batch size --
1024 pairs put one wavefront on every SIMD, 2048 two (python tools/dev/duplex_by_batch.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fast_ctc_decode_amd import _native as nat
import fast_ctc_decode_amd as fcd
from duplex_account import gen
T, w = 2000, 64
i = np.arange(T)
env = np.stack([np.maximum(0, i - w), np.minimum(T, i + w)], 1).astype(np.uint64)
h = nat.default_handle()
for B in (256, 512, 1024, 2048, 3072, 4096):
    x1, x2 = gen(B, T, 5, 4), gen(B, T, 5, 5)
    envs = torch.from_numpy(np.broadcast_to(env, (B, T, 2)).copy().view(np.int64)).cuda()
    out = []
    for mode, name in ((0, "logsumexp"), (1, "max")):
        ms = []
        for _ in range(3):
            r = fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, logadd_mode=mode)
            torch.cuda.synchronize()
            ms.append(h.last_kernel_ms())
        m = min(ms[1:])
        out.append("%s %.2f ms = %.0f pairs/s" % (name, m, B / m * 1e3))
    print("%5d pairs: %s" % (B, " | ".join(out)), flush=True)
    del x1, x2, envs
