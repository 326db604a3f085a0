#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r06l}; mkdir -p $O
python -c "import torch" 2>/dev/null
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print('%-10s %-7s value %.0f  ms_per_step %.4f  kernel_ms %.4f' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], r.get('kernel_ms') or 0))" "$1" "$2"; }
for rep in 1 2; do
  for d in ab_6d584a2 ab_43c577d .; do
    for o in stable pdq178; do
      (cd $d && FCD_TIE_ORDER=$o python bench.py --steps 20 --warmup 3 --no-viterbi --no-e2e --cpu-seconds 0.3 2>/dev/null | tail -1 | line $d $o) >> $O/orders.txt
    done
  done
done
cat $O/orders.txt
