#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-r06n}; O=gpurun_out/$TAG; mkdir -p $O
python -c "import torch" 2>/dev/null
./tools/microbench/f64lat > $O/f64lat.txt 2>&1; cat $O/f64lat.txt
timeout 900 python -m pytest tests/test_gpu_duplex.py -m gpu -x -q > $O/pytest_duplex.log 2>&1; tail -2 $O/pytest_duplex.log
bash tools/profile.sh ${TAG} duplex > $O/profile.log 2>&1
python tools/duplex_account.py > $O/duplex_account.jsonl 2>/dev/null; cut -c1-330 $O/duplex_account.jsonl
