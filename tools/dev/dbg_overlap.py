"""developer probe: two overlapping calls into the same result arrays"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import fast_ctc_decode_amd as fcd
from fast_ctc_decode_amd import _native as nat
from test_gpu_parity import gen_batch
xs = [gen_batch(900 + i, 9, 200 + 16 * i, 5) for i in range(5)]
h = nat.default_handle()
h.set_workspace_limit(6 << 20)
h.check(h.lib.fcd_debug_set_first_pass_divisor(h.ptr, 6))
serial = [fcd.beam_search_batch_raw(x, 32, 0.05, True, kernel=fcd.KERNEL_LANE) for x in xs]
xt = [torch.from_numpy(x).cuda() for x in xs]
B, w = 9, 264
def run(order, overlap, between=None):
    h.set_overlap(overlap)
    labels = torch.zeros((B, w), dtype=torch.uint8, device="cuda"); path = torch.zeros((B, w), dtype=torch.int32, device="cuda")
    out_len = torch.zeros(B, dtype=torch.int32, device="cuda"); status = torch.full((B,), -7, dtype=torch.int32, device="cuda")
    res = nat.Result(labels.data_ptr(), path.data_ptr(), None, out_len.data_ptr(), status.data_ptr(), w, None)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    for k in order:
        t = xt[k]; st_ = t.stride()
        b = nat.Batch(t.data_ptr(), t.shape[0], t.shape[1], 1, 5, st_[0], st_[1], 0, st_[2], None, nat.DTYPE_F32)
        h.check(h.lib.fcd_beam_search_dev(h.ptr, C.byref(b), 32, 0.05, 1, fcd.KERNEL_LANE, C.byref(res)))
        if between == "fcd": h.synchronize()
        if between == "torch": torch.cuda.synchronize()
        if between == "print": print("   after", k, out_len.cpu().numpy())
    h.overlap_join(); torch.cuda.synchronize()
    print(order, overlap, out_len.cpu().numpy(), status.cpu().numpy(), "want", serial[order[-1]].out_len)
run((4, 1), 0); run((4, 1), 3); run((4, 1), 3, "fcd"); run((4, 1), 3, "torch"); run((4, 1), 3, "print"); run((4, 4, 1), 3, "torch"); run((1, 4, 1), 3, "torch")
