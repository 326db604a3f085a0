#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06x; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlapping or two_pass" > $O/pytest_overlap.log 2>&1; tail -3 $O/pytest_overlap.log
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$name", round(d["value"]), round(d["ms_per_step"],3), "kernel_ms", round(r["kernel_ms"],2), "single", r.get("overlap") and round(r["overlap"]["single_launch_ms"],3))
except Exception as e: print("$name failed", e); print(open("$O/$name.err").read()[-800:])
PY
}
C="python bench.py --no-viterbi --no-e2e --cpu-seconds 1"
run c2_ov0 $C --overlap 0
run c2_ov2 $C --overlap 2
run c2_ov3 $C --overlap 3
run c2_ov4 $C --overlap 4
run c2_ov8 $C --overlap 8
run c2_ov4_s40 $C --overlap 4 --steps 40 --warmup 8
run c2_ov8_s40 $C --overlap 8 --steps 40 --warmup 8
run c4_ov4 $C --config 4 --overlap 4
run c4_ov0 $C --config 4 --overlap 0
FCD_TIE_ORDER=stable run c2_stable_ov4 $C --overlap 4
run c2_b16384_ov0 $C --overlap 0 --batch 16384
run c2_b16384_ov4 $C --overlap 4 --batch 16384
