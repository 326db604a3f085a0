#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r06q}; mkdir -p $O
python -c "import torch" 2>/dev/null
for rep in 1 2; do
  for v in w_before w_after; do
    python tools/dev/time_ragged.py ab_variants/$v.so 2>&1 | grep ragged >> $O/ragged.txt
    REPS=20 python tools/dev/time_variant.py ab_variants/$v.so 2>&1 | grep reference >> $O/ragged.txt
  done
done
cat $O/ragged.txt
