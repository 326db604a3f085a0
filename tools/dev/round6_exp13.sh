#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06v; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 300 python tools/dev/dbg_overlap.py 2>&1 | tail -12
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$name", round(d["value"]), round(d["ms_per_step"],2), "kernel_ms", round(r["kernel_ms"],2))
except Exception as e: print("$name failed", e); print(open("$O/$name.err").read()[-800:])
PY
}
C="python bench.py --config 3 --no-viterbi --no-e2e --cpu-seconds 1 --warmup 8"
FCD_OVERLAP_PRIORITY=high run hi_ov4 $C --overlap 4 --steps 16
FCD_OVERLAP_PRIORITY=low run lo_ov4 $C --overlap 4 --steps 16
FCD_OVERLAP_PRIORITY=high run hi_ov6 $C --overlap 6 --steps 18
GPU_MAX_HW_QUEUES=12 run q12_ov8 $C --overlap 8 --steps 24
GPU_MAX_HW_QUEUES=8 run q8_ov5 $C --overlap 5 --steps 20
GPU_MAX_HW_QUEUES=8 FCD_TIE_ORDER=stable run q8_stable_ov4 $C --overlap 4 --steps 16
GPU_MAX_HW_QUEUES=8 FCD_TIE_ORDER=stable run q8_stable_ov6 $C --overlap 6 --steps 18
