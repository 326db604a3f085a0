#!/bin/bash
# round 5, first GPU call: the new tie-order replay (pdq178_wave.h) on the hardware -- parity, its cycle account (probe and
# in place), kernel times by beam under both orders, the bench variants the occupancy fix is about, config 3 on 1/2/4 streams
set -u
T=${1:-r05a}
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/${T}_pytest_gpu.log 2>&1; tail -n 3 $O/${T}_pytest_gpu.log
python tools/dev/time_coop.py > $O/${T}_time_coop.txt 2>&1
python tools/dev/lane_tie_prof.py > $O/${T}_lane_tie_prof.txt 2>&1; cat $O/${T}_lane_tie_prof.txt
( for b in 5 8 12; do BEAM=$b REPS=5 python tools/dev/time_variant.py; done; BEAM=32 BATCH=8192 REPS=3 python tools/dev/time_variant.py ) > $O/${T}_by_beam.txt 2>&1; cat $O/${T}_by_beam.txt
python bench.py --no-e2e --no-viterbi --cpu-seconds 1 > $O/${T}_bench.json 2> $O/${T}_bench.err
( python bench.py --streams 2 --no-viterbi --no-e2e --cpu-seconds 1; python bench.py --batch 16384 --no-viterbi --no-e2e --cpu-seconds 1 ) > $O/${T}_bench_variants.txt 2> $O/${T}_bench_variants.err
for s in 1 2 4; do python bench.py --config 3 --streams $s --no-viterbi --steps 8 --warmup 4 --cpu-seconds 1; done > $O/${T}_bench_config3_streams.txt 2> $O/${T}_bench_config3_streams.err
FCD_TIE_ORDER=stable bash tools/profile_sq.sh ${T}_lane_stable beam32 > $O/${T}_sq_lane.log 2>&1
for f in $O/${T}_bench.json $O/${T}_bench_variants.txt $O/${T}_bench_config3_streams.txt; do python - $f <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    print(sys.argv[1].split('/')[-1], d.get("config", {}).get("workload", "")[:40], "value %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "kernel_ms", d.get("roofline", {}).get("kernel_ms"), "streams", d.get("config", {}).get("streams"))
PY
done
