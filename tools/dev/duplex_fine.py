"""developer scratch: the slot-resident duplex kernel's step cut into 16 consecutive intervals (a -DFCD_SLOTS_FINE build of
duplex_slots.hip: EXTRA=-DFCD_SLOTS_FINE tools/dev/mk_variant.sh fine duplex_slots.hip), cycles per step of BASELINE config 5
    python tools/dev/duplex_fine.py ab_variants/fine.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fast_ctc_decode_amd import _native as nat
nat.LIB_PATH = os.path.abspath(sys.argv[1])
import fast_ctc_decode_amd as fcd
from duplex_account import gen
NAMES = ["envelope + tile", "sort, parents, extension prologue", "rescans", "extension rows", "root staging", "expansion: entry fields, candidates",
         "expansion: probabilities", "new-node slots", "builds", "probability, keys, exact rank", "ties", "survivors, returning loads issued",
         "eviction lists, parents' bounds, rank lanes", "returning nodes landed", "evictions", "free list, end of step"]
B, T, w = 1024, 2000, 64
x1, x2 = gen(B, T, 5, 4), gen(B, T, 5, 5)
i = np.arange(T)
env = np.stack([np.maximum(0, i - w), np.minimum(T, i + w)], 1).astype(np.uint64)
envs = torch.from_numpy(np.broadcast_to(env, (B, T, 2)).copy().view(np.int64)).cuda()
h = nat.default_handle()
for mode, name in ((0, "logsumexp"), (1, "max")):
    prof = torch.zeros((B, 32), dtype=torch.int32, device="cuda")
    h.check(h.lib.fcd_debug_set_duplex_profile(h.ptr, C.c_void_p(prof.data_ptr())))
    fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, logadd_mode=mode)
    torch.cuda.synchronize()
    h.check(h.lib.fcd_debug_set_duplex_profile(h.ptr, None))
    a = prof.cpu().numpy().astype(np.float64)
    per = (a[:, 16:32] * 64.0 / a[:, 7:8]).mean(0)
    print(name, "total %.0f cycles per step" % per.sum())
    for k in range(16):
        print("   %2d %-48s %8.1f" % (k, NAMES[k], per[k]))
