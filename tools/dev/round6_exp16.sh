#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06y; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
for c in 2 3; do
python bench.py --config $c --force-dist --no-viterbi --no-e2e --cpu-seconds 1 > $O/dist1_c$c.json 2> $O/dist1_c$c.err; tail -1 $O/dist1_c$c.json | cut -c1-160
python bench.py --config $c --force-dist --no-overlap --no-viterbi --no-e2e --cpu-seconds 1 > $O/dist1_nooverlap_c$c.json 2> $O/dist1_nooverlap_c$c.err; tail -1 $O/dist1_nooverlap_c$c.json | cut -c1-160
python bench.py --config $c --force-dist --overlap 0 --no-viterbi --no-e2e --cpu-seconds 1 > $O/dist1_ov0_c$c.json 2> $O/dist1_ov0_c$c.err; tail -1 $O/dist1_ov0_c$c.json | cut -c1-160
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 > $O/torchrun1.json 2> $O/torchrun1.err; tail -1 $O/torchrun1.json | cut -c1-160
