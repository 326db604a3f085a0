"""The host batch path with the reads held as float16 against float32: the PCIe legs, the device search, and the
compiled beam_search_batch under a few pipeline settings (lanes x chunk size).

    python tools/probe_f16_e2e.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fast_ctc_decode_amd as fcd
from fast_ctc_decode_amd import api, _native as nat
cm = api._compiled()
rng = np.random.default_rng(1)
B=4096
x = rng.random((B*4000, 5), dtype=np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True); x = x.reshape(B,4000,5)
xh = x.astype(np.float16)
def best(fn, n=3):
    fn(); ts=[]
    for _ in range(n):
        t0=time.perf_counter(); r=fn(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0); r=None
    return min(ts)*1e3
xd = torch.empty((B,4000,5), dtype=torch.float16, device="cuda")
xt = torch.from_numpy(xh)
print("upload f16 pageable %.2f ms" % best(lambda: xd.copy_(xt)))
xd32 = torch.empty((B,4000,5), dtype=torch.float32, device="cuda"); xt32 = torch.from_numpy(x)
print("upload f32 pageable %.2f ms" % best(lambda: xd32.copy_(xt32)))
print("device search f16 %.2f ms, f32 %.2f ms" % (best(lambda: fcd.beam_search_batch_raw(xd,5,0.1)), best(lambda: fcd.beam_search_batch_raw(xd32,5,0.1))))
for lanes, chunk in ((0,0),(4,1024),(4,512),(8,512),(3,1408),(2,2048)):
    cm._set_host_pipeline(lanes, chunk, -1)
    print("lanes %d chunk %d: no paths f32 %.2f ms  f16 %.2f ms | array f32 %.2f f16 %.2f" % (lanes, chunk,
          best(lambda: cm.beam_search_batch(x,"NACGT",5,0.1,paths=None)), best(lambda: cm.beam_search_batch(xh,"NACGT",5,0.1,paths=None)),
          best(lambda: cm.beam_search_batch(x,"NACGT",5,0.1,paths="array")), best(lambda: cm.beam_search_batch(xh,"NACGT",5,0.1,paths="array"))), flush=True)
