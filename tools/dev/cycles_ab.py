"""developer scratch: tools/cycle_account.py's per-block account at 4096 reads, for a given build of the library, under
both tie orders (python tools/dev/cycles_ab.py [LIB])"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tools"))
from fast_ctc_decode_amd import _native as nat
if len(sys.argv) > 1:
    nat.LIB_PATH = os.path.abspath(sys.argv[1])
import fast_ctc_decode_amd as fcd
import cycle_account
for order in ("stable", "pdq178"):
    fcd.set_tie_order(order)
    print(os.path.basename(nat.LIB_PATH), order, flush=True)
    sys.argv = ["x", "4096"]
    cycle_account.main()
