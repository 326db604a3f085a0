#!/bin/bash
# round 6, GPU call 3+: A/B of library variants (ab_variants/*.so, tools/dev/mk_variant.sh) on one box: the config-2 kernel
# under both tie orders, alternating (tools/dev/time_variant.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r06c}; mkdir -p $O
python -c "import torch" 2>/dev/null
for rep in 1 2; do
  for v in ${VARIANTS:-base vA}; do
    REPS=20 python tools/dev/time_variant.py ab_variants/$v.so 2>&1 | grep -v "^$" | grep reference >> $O/variants.txt
  done
done
cat $O/variants.txt
