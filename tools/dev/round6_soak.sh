#!/bin/bash
# differential soaks of the round's final tree against the oracle, side by side (beam_soak: half the seeds through the
# device-side slab pool under a workspace limit, a third with the calls on internal streams)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
python -c "import torch" 2>/dev/null
FCD_SOAK_SECONDS=${1:-420} python tools/beam_soak.py ${3:-9700000} 100000000 > $O/r06z_soak_beam.log 2>&1 &
FCD_SOAK_SECONDS=${1:-420} python tools/duplex_soak.py ${4:-9800000} 100000000 > $O/r06z_soak_duplex.log 2>&1 &
FCD_SOAK_SECONDS=${2:-200} python tools/hostjob_soak.py ${5:-9900000} 100000000 > $O/r06z_soak_hostjob.log 2>&1 &
for s in 11 12 13; do timeout 300 python tools/overlap_soak.py 1000 $s 2>&1 | grep -v amdgpu | tail -1; done > $O/r06z_soak_overlap.log 2>&1
wait
tail -n 3 $O/r06z_soak_beam.log $O/r06z_soak_duplex.log $O/r06z_soak_hostjob.log $O/r06z_soak_overlap.log
