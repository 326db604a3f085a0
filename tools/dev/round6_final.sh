#!/bin/bash
# round 6: the evidence set on the round's kernels in ONE GPU call -- the -m gpu suite, smoke(), tools/round_profiles.sh
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-r06z}; O=gpurun_out; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q > $O/${TAG}_pytest_gpu.log 2>&1; tail -2 $O/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; tail -1 $O/${TAG}_smoke.log
bash tools/round_profiles.sh $TAG > $O/${TAG}_round_profiles.log 2>&1; tail -12 $O/${TAG}_round_profiles.log | cut -c1-300
python tools/dev/duplex_by_batch.py 2>&1 | grep -v amdgpu > $O/${TAG}_duplex_pairs_by_batch.txt; cat $O/${TAG}_duplex_pairs_by_batch.txt
