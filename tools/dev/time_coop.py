"""developer scratch: latency of one cooperative quicksort replay (pdq178_wave.h) -- one wavefront per SIMD"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
from fast_ctc_decode_amd import _native as nat
from test_pdq178 import orderable

if os.environ.get("FCD_LIB"):  # an A/B candidate of tools/dev/mk_variant.sh
    nat.LIB_PATH = os.path.abspath(os.environ["FCD_LIB"])
lib = nat.load()
h = nat.default_handle(0)
h.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)


def run(n, planes, keep, pairs=1024, serial=False):
    lists = []
    for _ in range(2 * pairs):
        vals = rng.random(6, dtype=np.float32)
        p = np.where(rng.random(n) < 0.3, vals[rng.integers(0, 6, n)], rng.random(n, dtype=np.float32)).astype(np.float32)
        lists.append(p)
    buf = np.zeros((len(lists), n), np.uint64)
    for i, p in enumerate(lists):
        buf[i] = (orderable(p) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    lens = np.full(len(lists), n, np.int32)
    d0 = torch.from_numpy(buf.view(np.int64)).cuda()
    dl = torch.from_numpy(lens).cuda()
    best = 1e9
    for _ in range(5):
        d = d0.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if serial:
            rc = lib.fcd_debug_pdq178_sort_dev(h.ptr, d.data_ptr(), len(lists), n, dl.data_ptr())
        else:
            rc = lib.fcd_debug_pdq178_coop_sort_dev(h.ptr, d.data_ptr(), len(lists), n, dl.data_ptr(), planes, keep)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0
        best = min(best, e0.elapsed_time(e1))
    print("n=%d planes=%d keep=%d %s: %.1f us per launch (%d lists, one wavefront each)" %
          (n, planes, keep, "serial" if serial else "coop", best * 1e3, pairs), flush=True)
    if not serial:
        import ctypes as C
        cyc = (C.c_uint64 * 16)()
        lib.fcd_debug_pdq178_coop_profile(h.ptr, cyc, 1)
        calls = max(int(cyc[11]), 1)
        names = ["setup", "pivot", "swap+mode", "classify", "scans+tables", "moves", "leftovers+children", "next", "exit", "leaves"]
        print("    cycles per call: " + ", ".join("%s %.0f" % (nm, cyc[i] / calls) for i, nm in enumerate(names)) +
              "; segments per call %.2f; total %.0f" % (cyc[10] / calls, sum(cyc[:10]) / calls), flush=True)


if os.environ.get("REG_ONLY"):  # (a -DFCD_REG_PROF build: slots 1 .. 6 are the phases of a partition in registers)
    for n in (25, 40, 64):
        for keep in (1 << 20, 32):
            run(n, 1, keep)
    sys.exit(0)
for n, planes in ((25, 1), (128, 3), (64, 3), (160, 3), (130, 5)):
    for keep in (1 << 20, 32, 5):
        run(n, planes, keep)
run(128, 3, 0, serial=True)
run(25, 1, 0, serial=True)
