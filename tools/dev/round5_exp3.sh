#!/bin/bash
# round 5: does the issue priority of tied wavefronts let launches on several streams overlap their tails?
set -u
T=${1:-r05c}
O=gpurun_out; mkdir -p $O
for s in 1 2 3 4 6; do python bench.py --config 3 --streams $s --no-viterbi --steps 12 --warmup 6 --cpu-seconds 0.5 2>> $O/${T}_streams.err; done > $O/${T}_bench_config3_streams.txt
python - $O/${T}_bench_config3_streams.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    print("streams", d["config"].get("streams"), "value %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "kernel_ms %.2f" % d["roofline"]["kernel_ms"])
PY
tail -n 3 $O/${T}_streams.err
