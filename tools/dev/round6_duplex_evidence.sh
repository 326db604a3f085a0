#!/bin/bash
# round 6: evidence set of the slot-resident duplex kernel -- soaks against the oracle (many processes side by side: the
# CPU oracle is the slow half), rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE of config 5, pairs/s by batch size.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-r06m}; O=gpurun_out/$TAG; mkdir -p $O
python -c "import torch" 2>/dev/null
SECS=${SECS:-300}
for k in 1 2 3 4 5 6 7 8 9 10 11 12; do
  FCD_SOAK_SECONDS=$SECS timeout -k 10 $((SECS + 120)) python tools/duplex_soak.py $((3000000 + k * 100000)) 100000 > $O/soak_$k.log 2>&1 &
done
for k in 1 2 3 4; do
  FCD_SOAK_SECONDS=$SECS timeout -k 10 $((SECS + 120)) python tools/duplex_soak.py --long $((4000000 + k * 100000)) 100000 > $O/soak_long_$k.log 2>&1 &
done
wait
grep -h "MISMATCH" $O/soak_*.log | head -20
tail -q -n1 $O/soak_*.log > $O/duplex_soak.txt; cat $O/duplex_soak.txt
bash tools/profile.sh ${TAG} duplex > $O/profile.log 2>&1
python tools/dev/duplex_by_batch.py > $O/duplex_pairs_by_batch.txt 2>&1; cat $O/duplex_pairs_by_batch.txt
python tools/duplex_account.py > $O/duplex_account.jsonl 2>/dev/null; cut -c1-330 $O/duplex_account.jsonl
