#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06z; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_duplex.py tests/test_gpu_parity.py -x -q -m gpu -k "overlapping" > $O/pytest_overlap.log 2>&1; tail -2 $O/pytest_overlap.log
python tools/duplex_overlap.py 2>&1 | grep -v amdgpu.ids | tee $O/duplex_overlap.txt
python tools/tie_order_by_beam.py 2>&1 | grep -v amdgpu.ids | tee $O/tie_order_by_beam.txt
for c in 2 3; do
python bench.py --config $c --force-dist --no-viterbi --no-e2e --cpu-seconds 1 > $O/dist1_c$c.json 2> $O/dist1_c$c.err; tail -1 $O/dist1_c$c.json | cut -c1-160
done
