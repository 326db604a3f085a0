#!/bin/bash
# round 6, GPU call 1: (a) headline A/B -- round 3's tree (git worktree in ab_r03/, built in place) next to HEAD, alternating,
# same box, both tie orders for HEAD; (b) this box's baseline of the duplex account and the bench line before round-6 changes.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06a; mkdir -p $O
python -c "import torch" 2>/dev/null
for rep in 1 2 3; do
  (cd ab_r03 && python bench.py --steps 20 --warmup 3 --no-viterbi --no-e2e --cpu-seconds 0.5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('r03  stable  value %.0f  ms_per_step %.4f  kernel_ms %s' % (d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel_ms')))") >> $O/headline_ab.txt
  FCD_TIE_ORDER=stable python bench.py --steps 20 --warmup 3 --no-viterbi --no-e2e --cpu-seconds 0.5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('HEAD stable  value %.0f  ms_per_step %.4f  kernel_ms %s' % (d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel_ms')))" >> $O/headline_ab.txt
  python bench.py --steps 20 --warmup 3 --no-viterbi --no-e2e --cpu-seconds 0.5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('HEAD pdq178  value %.0f  ms_per_step %.4f  kernel_ms %s' % (d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel_ms')))" >> $O/headline_ab.txt
done
cat $O/headline_ab.txt
python tools/duplex_account.py > $O/duplex_account_before.jsonl 2> $O/duplex_account_before.err
cat $O/duplex_account_before.jsonl
