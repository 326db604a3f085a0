#!/bin/bash
# round 6, GPU call: the slot-resident duplex kernel -- parity suite + cycle account, with and without the L2 prefetch of stale candidates
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r06g}; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_duplex.py -m gpu -x -q > $O/pytest_duplex.log 2>&1; tail -3 $O/pytest_duplex.log
timeout 300 python tools/duplex_account.py > $O/duplex_account.jsonl 2> $O/duplex_account.err; cat $O/duplex_account.jsonl; tail -3 $O/duplex_account.err
FCD_DUPLEX_PREFETCH=0 timeout 300 python tools/duplex_account.py > $O/duplex_account_noprefetch.jsonl 2>/dev/null; cat $O/duplex_account_noprefetch.jsonl
