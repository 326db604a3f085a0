#!/bin/bash
# fcd_set_overlap + the device-side slab pool: tests first, then config 3 by the number of internal streams
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06u; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lane or overlapping" > $O/pytest_lane.log 2>&1; tail -3 $O/pytest_lane.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_tieorder.py -x -q -m gpu > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$name", round(d["value"]), round(d["ms_per_step"],2), "kernel_ms", round(r["kernel_ms"],2), "overlap", r.get("overlap") and (r["overlap"]["streams"], round(r["overlap"]["single_launch_ms"],2)), "mism", (d.get("cpu_baseline") or {}).get("mismatches"))
except Exception as e: print("$name failed", e); print(open("$O/$name.err").read()[-800:])
PY
}
C="python bench.py --config 3 --no-viterbi --no-e2e --cpu-seconds 1 --warmup 8"
run ov0 $C --overlap 0 --steps 5
run ov2 $C --overlap 2 --steps 12
run ov3 $C --overlap 3 --steps 12
run ov4 $C --overlap 4 --steps 16
run ov6 $C --overlap 6 --steps 18
GPU_MAX_HW_QUEUES=8 run q8_ov4 $C --overlap 4 --steps 16
GPU_MAX_HW_QUEUES=8 run q8_ov6 $C --overlap 6 --steps 18
FCD_TIE_ORDER=stable run stable_ov0 $C --overlap 0 --steps 5
FCD_TIE_ORDER=stable run stable_ov4 $C --overlap 4 --steps 16
rocm-smi --showmemuse 2>/dev/null | head -8
