bash tools/dev/probe_sort_sq.sh 2>&1 | tail -3
bash tools/dev/round5_exp2.sh r05e
