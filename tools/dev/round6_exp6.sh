#!/bin/bash
# round 6, GPU call: duplex kernel variants (ab_variants/*.so) on one box: config-5 kernel time, both flavours (tools/dev/time_duplex.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r06i}; mkdir -p $O
python -c "import torch" 2>/dev/null
for rep in 1 2; do
  for v in ${VARIANTS}; do
    FCD_DUPLEX_PREFETCH=${PREFETCH:-0} python tools/dev/time_duplex.py ab_variants/$v.so 2>&1 | grep -v "^$" | grep logsumexp >> $O/duplex_variants.txt
  done
done
cat $O/duplex_variants.txt
