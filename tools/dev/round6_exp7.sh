#!/bin/bash
# round 6: the whole -m gpu suite (with the r06 additions) + headline alignment variants
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r06k}; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
for rep in 1 2; do
  for v in ${VARIANTS}; do
    REPS=20 python tools/dev/time_variant.py ab_variants/$v.so 2>&1 | grep reference >> $O/variants.txt
  done
done
cat $O/variants.txt
