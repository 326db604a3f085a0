#!/bin/bash
# the bench lines once more, with the round's counter summaries in place (roofline.traffic resolves against profiles/TAG_pmc_summary.json)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-r06z}; O=gpurun_out
python -c "import torch" 2>/dev/null
python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err
python bench.py --config 3 --no-viterbi > $O/${TAG}_bench_config3.json 2> $O/${TAG}_bench_config3.err
python bench.py --config 4 --no-viterbi > $O/${TAG}_bench_config4.json 2> $O/${TAG}_bench_config4.err
python bench.py --config 5 > $O/${TAG}_bench_config5.json 2> $O/${TAG}_bench_config5.err
python bench.py --config 5 --mode max > $O/${TAG}_bench_config5_max.json 2>> $O/${TAG}_bench_config5.err
python bench.py --overlap 0 --no-e2e --cpu-seconds 1 > $O/${TAG}_bench_line_overlap0.json 2> $O/${TAG}_bench_overlap0.err
python bench.py --config 3 --overlap 0 --no-viterbi --steps 5 --cpu-seconds 1 > $O/${TAG}_bench_config3_overlap0.json 2>> $O/${TAG}_bench_overlap0.err
FCD_TIE_ORDER=stable python bench.py --no-viterbi --no-e2e --cpu-seconds 1 > $O/${TAG}_bench_line_stable_order.json 2> $O/${TAG}_bench_stable.err
FCD_TIE_ORDER=stable python bench.py --config 3 --no-viterbi --cpu-seconds 1 > $O/${TAG}_bench_config3_stable_order.json 2>> $O/${TAG}_bench_stable.err
for f in bench_line bench_config3 bench_config4 bench_config5 bench_config5_max bench_line_overlap0 bench_config3_overlap0 bench_line_stable_order bench_config3_stable_order; do tail -1 $O/${TAG}_$f.json | cut -c1-200; done
