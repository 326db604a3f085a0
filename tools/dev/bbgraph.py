"""developer aid: basic blocks of a kernel's main loop (python tools/dev/bbgraph.py OBJ KERNEL-SUBSTRING [LOOP_SIZE]):
per block its address, instruction count, a few landmark opcodes and where its terminator goes."""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "loop_hist.py")).read()
exec(src.split("text = disasm(sys.argv[1])")[0])
if len(sys.argv) > 3:
    os.environ["LOOP_SIZE"] = sys.argv[3]
body = main_loop(disasm(sys.argv[1]), sys.argv[2])
addrs = [a for a, _, _ in body]
lo, hi = addrs[0], addrs[-1]
targets = {}
for a, op, ar in body:
    if op.startswith("s_cbranch") or op == "s_branch":
        off = int(ar.split()[-1])
        if off >= 32768:
            off -= 65536
        targets[a] = a + 4 + off * 4
leaders = {lo} | {t for t in targets.values() if lo <= t <= hi}
for i, (a, op, ar) in enumerate(body):
    if a in targets and i + 1 < len(body):
        leaders.add(body[i + 1][0])
blocks, cur = [], []
for ins in body:
    if ins[0] in leaders and cur:
        blocks.append(cur)
        cur = []
    cur.append(ins)
blocks.append(cur)
MARK = ("global_", "ds_bpermute", "ds_permute", "v_rcp", "v_div_fixup", "ds_write", "ds_read", "s_waitcnt", "v_readlane", "s_sleep")
for b in blocks:
    a, op, ar = b[-1]
    if a in targets:
        t = targets[a]
        term = "%s -> %x%s" % (op, t, "" if lo <= t <= hi else " (out)")
    else:
        term = "falls through"
    marks = {}
    for _, o, _ in b:
        for m in MARK:
            if o.startswith(m):
                marks[m] = marks.get(m, 0) + 1
    print("%x  %3d  %-34s %s" % (b[0][0], len(b), term, " ".join("%s%d" % (k.rstrip("_"), v) for k, v in marks.items())))
