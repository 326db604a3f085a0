"""developer scratch: where a tie-flagged step of the wide-beam kernel spends its cycles, measured IN PLACE (a
-DFCD_LANE_TIE_PROF build of beam_lane.hip: tools/dev/lane_tie_prof.sh) on the BASELINE config-3 shard"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fast_ctc_decode_amd import _native as nat
nat.LIB_PATH = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "tools/dev/_build/libfcd_hip_prof.so")
import fast_ctc_decode_amd as fcd
import bench
BEAM = int(os.environ.get("BEAM", "32"))
BATCH = int(os.environ.get("BATCH", "8192"))
x = torch.from_numpy(bench.make_batch(2, BATCH)).cuda()
fcd.set_tie_order("pdq178")
r = fcd.beam_search_batch_raw(x, BEAM, 0.1, True)
torch.cuda.synchronize()
h = r._handle
lib = nat.load()
cyc = (C.c_uint64 * 16)()
lib.fcd_debug_pdq178_coop_profile(h.ptr, cyc, 3)  # read + reset the in-place counters
h.timing_reset()
r = fcd.beam_search_batch_raw(x, BEAM, 0.1, True)
torch.cuda.synchronize()
print("kernel %.2f ms (with stamps)" % h.timing_mean_ms()[0])
lib.fcd_debug_pdq178_coop_profile(h.ptr, cyc, 3)
steps = max(int(cyc[15]), 1)
calls = max(int(cyc[11]), 1)
print("tied steps %d, replays %d; per tied step: list %.0f cycles, replay %.0f, hand-back %.0f" %
      (steps, calls, cyc[12] / steps, cyc[13] / steps, cyc[14] / steps))
names = ["setup", "pivot", "swap+mode", "classify", "scans+tables", "moves", "leftovers+children", "next", "exit", "leaves"]
print("per replay: " + ", ".join("%s %.0f" % (nm, cyc[i] / calls) for i, nm in enumerate(names)) +
      "; segments %.2f; total %.0f" % (cyc[10] / calls, sum(cyc[:10]) / calls))
