"""developer scratch: a few launches of the quicksort probe kernel (tools/dev/probe_sort_sq.sh counts its instructions)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
from fast_ctc_decode_amd import _native as nat
from test_pdq178 import orderable
lib = nat.load()
h = nat.default_handle(0)
h.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
n, planes, keep = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L = 4096
buf = np.zeros((L, n), np.uint64)
for i in range(L):
    vals = rng.random(6, dtype=np.float32)
    p = np.where(rng.random(n) < 0.3, vals[rng.integers(0, 6, n)], rng.random(n, dtype=np.float32)).astype(np.float32)
    buf[i] = (orderable(p) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
d0 = torch.from_numpy(buf.view(np.int64)).cuda()
dl = torch.from_numpy(np.full(L, n, np.int32)).cuda()
for _ in range(2):
    d = d0.clone()
    assert lib.fcd_debug_pdq178_coop_sort_dev(h.ptr, d.data_ptr(), L, n, dl.data_ptr(), planes, keep) == 0
torch.cuda.synchronize()
print("ok", n, planes, keep)
