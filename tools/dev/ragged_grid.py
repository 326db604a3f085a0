import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import fast_ctc_decode_amd as fcd
from fast_ctc_decode_amd import api
cm = api._compiled()
B = 4096
rng = np.random.default_rng(1)
x = rng.random((B * 4000, 5), dtype=np.float32); x /= np.linalg.norm(x, ord=2, axis=1, keepdims=True); x = x.reshape(B, 4000, 5)
rows = rng.integers(2000, 4001, B)
reads = [np.ascontiguousarray(x[i, :rows[i]]) for i in range(B)]
def best(fn, n=3):
    fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
for lanes, chunk in ((3, 0), (3, 512), (4, 512), (4, 256), (6, 512), (6, 256), (8, 256), (8, 512), (6, 704), (12, 352)):
    cm._set_host_pipeline(lanes, chunk, -1)
    ms = best(lambda: cm.beam_search_batch(reads, "NACGT", 5, 0.1, paths="array"))
    print("lanes=%d chunk=%d: %.2f ms = %.0fk reads/s" % (lanes, chunk, ms, B / ms), flush=True)
