#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06w; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$name", round(d["value"]), round(d["ms_per_step"],2), "kernel_ms", round(r["kernel_ms"],2), "single", r.get("overlap") and round(r["overlap"]["single_launch_ms"],2), (d.get("cpu_baseline") or {}).get("outputs_match"))
except Exception as e: print("$name failed", e); print(open("$O/$name.err").read()[-800:])
PY
}
C="python bench.py --config 3 --no-viterbi --no-e2e --cpu-seconds 1 --warmup 8"
run ov3 $C --overlap 3 --steps 12
run ov4 $C --overlap 4 --steps 16
run ov5 $C --overlap 5 --steps 20
run ov6 $C --overlap 6 --steps 18
run ov8 $C --overlap 8 --steps 24
FCD_TIE_ORDER=stable run stable_ov4 $C --overlap 4 --steps 16
FCD_TIE_ORDER=stable run stable_ov8 $C --overlap 8 --steps 24
python bench.py --config 3 --no-viterbi --steps 10 > $O/bench_config3.json 2> $O/bench_config3.err; tail -1 $O/bench_config3.json | cut -c1-300
python bench.py > $O/bench_line.json 2> $O/bench_line.err; tail -1 $O/bench_line.json | cut -c1-200
