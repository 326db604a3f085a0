#!/bin/bash
# round 5: the replay's cycle account (probe and in place) and kernel times by beam under both orders
set -u
T=${1:-r05b}
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_tieorder.py -x -q > $O/${T}_pytest_tie.log 2>&1; tail -n 2 $O/${T}_pytest_tie.log
python tools/dev/time_coop.py 2>&1 | grep -v amdgpu.ids > $O/${T}_time_coop.txt
python tools/dev/lane_tie_prof.py 2>&1 | grep -v amdgpu.ids > $O/${T}_lane_tie_prof.txt; cat $O/${T}_lane_tie_prof.txt
( for b in 5 8 12; do BEAM=$b REPS=5 python tools/dev/time_variant.py; done; BEAM=32 BATCH=8192 REPS=3 python tools/dev/time_variant.py ) 2>&1 | grep -v amdgpu.ids > $O/${T}_by_beam.txt; cat $O/${T}_by_beam.txt
