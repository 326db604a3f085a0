"""crf_greedy_search on BASELINE config 4's shape (4096 reads x 4000 x 4 states x 5): kernel time by HIP events and the
HBM fraction on algorithmic bytes.   python tools/probe_greedy.py"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fast_ctc_decode_amd as fcd
g = torch.Generator(device="cuda"); g.manual_seed(3)
B=4096
x = torch.rand((B, 4000, 4, 5), generator=g, device="cuda")
init = torch.rand((B, 4), generator=g, device="cuda")
for _ in range(3): r = fcd.crf_greedy_search_batch_raw(x, init)
torch.cuda.synchronize()
h = r._handle; h.timing_reset()
for _ in range(10): r = fcd.crf_greedy_search_batch_raw(x, init)
torch.cuda.synchronize()
ms,_ = h.timing_mean_ms()
L = float(r.out_len.float().mean())
byt = B*(4000*4*5*4 + 5*L)
print("crf_greedy 4096x4000x4x5: %.3f ms, %.2f TB/s algorithmic (%.0f %% of the 8 TB/s peak), mean labels %.0f" % (ms, byt/ms/1e9, byt/ms/1e9/8*100, L))
