#!/bin/bash
# how far do sub-launches on several streams hide config 3's stragglers? (one handle + arena per stream: the existing --streams)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06t; mkdir -p $O
python -c "import torch" 2>/dev/null
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"]), round(d["ms_per_step"],2), d["config"].get("global_batch"), d.get("streams") or d["config"].get("streams"))
except Exception as e: print("$name failed", e)
PY
}
C="python bench.py --config 3 --no-viterbi --no-e2e --cpu-seconds 1"
run s3_b8192 $C --streams 3 --steps 9 --warmup 6
run s4_b4096 $C --streams 4 --batch 4096 --steps 16 --warmup 8
run s6_b4096 $C --streams 6 --batch 4096 --steps 18 --warmup 12
run s8_b2048 $C --streams 8 --batch 2048 --steps 32 --warmup 16
GPU_MAX_HW_QUEUES=8 run q8_s8_b2048 $C --streams 8 --batch 2048 --steps 32 --warmup 16
GPU_MAX_HW_QUEUES=8 run q8_s6_b4096 $C --streams 6 --batch 4096 --steps 18 --warmup 12
run s4_b8192 $C --streams 4 --steps 12 --warmup 8
GPU_MAX_HW_QUEUES=8 run q8_s4_b4096 $C --streams 4 --batch 4096 --steps 16 --warmup 8
