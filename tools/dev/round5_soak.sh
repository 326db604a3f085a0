#!/bin/bash
# differential soaks of the round's final kernels against the oracle, default tie order on both sides, side by side
set -u
O=gpurun_out; mkdir -p $O
FCD_SOAK_SECONDS=${1:-420} python tools/beam_soak.py ${3:-8500000} 100000000 > $O/r05i_soak_beam.log 2>&1 &
FCD_SOAK_SECONDS=${1:-420} python tools/duplex_soak.py ${4:-8600000} 100000000 > $O/r05i_soak_duplex.log 2>&1 &
FCD_SOAK_SECONDS=${2:-200} python tools/hostjob_soak.py ${5:-8800000} 100000000 > $O/r05i_soak_hostjob.log 2>&1 &
# the tie-order tests once more, and the replay against the committed vectors ON THE GPU
python - > $O/r05i_vectors_gpu.log 2>&1 <<'PY'
import json, os, sys
sys.path.insert(0, "tests")
import numpy as np, torch
from fast_ctc_decode_amd import _native as nat
from test_pdq178 import device_sort, device_coop_sort
doc = json.load(open("tools/verify/pdq178_vectors.json"))
lists = [np.array(c["bits"], np.uint32).view(np.float32) for c in doc["cases"]]
perms = [np.array(c["perm"], np.int64) for c in doc["cases"]]
class Dev:
    def __init__(self, a):
        self.t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda(); self.ptr = self.t.data_ptr()
h = nat.default_handle(0); h.set_stream(torch.cuda.current_stream().cuda_stream)
back = lambda d, shape, dt: d.t.cpu().numpy().view(dt).reshape(shape)
bad = 0
out, lens = device_sort(nat.load(), h, lists, Dev, back)
bad += sum(not np.array_equal((out[i, :lens[i]] & np.uint64(0xFFFFFFFF)).astype(np.int64), p) for i, p in enumerate(perms))
out, lens = device_coop_sort(nat.load(), h, lists, 8, Dev, back)
bad += sum(not np.array_equal((out[i, :lens[i]] & np.uint64(0xFFFFFFFF)).astype(np.int64), p) for i, p in enumerate(perms))
print("tools/verify/pdq178_vectors.json on the GPU: %d lists through pdq178.h (serial) and pdq178_wave.h + pdq178_reg.h: %d differ from the file" % (len(lists), bad))
PY
wait
tail -n 2 $O/r05i_soak_beam.log $O/r05i_soak_duplex.log $O/r05i_soak_hostjob.log $O/r05i_vectors_gpu.log
