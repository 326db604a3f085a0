#!/bin/bash
# round 6, GPU call 2: which commit slowed the STABLE-order headline kernel between round 3 (4.14 ms) and HEAD (4.27 ms)?
# Historical trees (git worktrees built in place under ab_<sha>/) run their own bench.py on one box, alternating.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06b; mkdir -p $O
python -c "import torch" 2>/dev/null
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print('%-10s %-7s value %.0f  ms_per_step %.4f  kernel_ms %.4f' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], r.get('kernel_ms') or 0))" "$1" "$2"; }
for rep in 1 2; do
  for d in ab_r03 ab_6d584a2 ab_fdfa862 ab_3dc1757 ab_83de397 ab_43c577d ab_401983d .; do
    [ -d $d ] || continue
    (cd $d && FCD_TIE_ORDER=stable python bench.py --steps 20 --warmup 3 --no-viterbi --no-e2e --cpu-seconds 0.3 2>/dev/null | tail -1 | line $d stable) >> $O/bisect.txt
  done
done
cat $O/bisect.txt
