"""tools/loop_hist.py OBJECT.o KERNEL-SUBSTRING [KERNEL-SUBSTRING-2]
Instruction histogram of the longest backward-branch loop of a kernel (its main time loop); with two kernels, the
difference of the two histograms.  Developer aid for comparing template instantiations."""
import collections, re, subprocess, sys, tempfile, os
B = "/opt/rocm/lib/llvm/bin"


def disasm(obj):
    t = tempfile.mkdtemp()
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, t + "/f.bin"])
    subprocess.check_call([B + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + t + "/f.bin", "--output=" + t + "/d.co", "--unbundle"])
    return subprocess.check_output([B + "/llvm-objdump", "-d", t + "/d.co"]).decode()


def main_loop(text, pat):
    m = [x for x in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^\n|\Z)", text, re.S | re.M) if pat in x.group(1)]
    assert len(m) == 1, [x.group(1) for x in m]
    ins = []
    for l in m[0].group(2).splitlines():
        mm = re.match(r"\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):", l)
        if mm:
            ins.append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
    loops = []
    for i, (addr, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            off = int(args.split()[-1])
            if off >= 32768:
                tgt = addr + 4 + (off - 65536) * 4
                loops.append((tgt, addr))
    # the time loop proper: the backward branch whose body is closest to `want` instructions (default: the longest)
    def size(lp):
        return sum(1 for a, _, _ in ins if lp[0] <= a <= lp[1])
    if os.environ.get("LOOP_LIST"):
        for lp in sorted(set(loops), key=size, reverse=True)[:12]:
            inside = [(a, op) for a, op, _ in ins if lp[0] <= a <= lp[1]]
            print("   loop %6x..%6x  %5d instructions, %3d scratch, %2d calls, %3d v_readlane/writelane, %3d global stores" % (
                lp[0], lp[1], len(inside), sum(op.startswith("scratch_") for _, op in inside),
                sum(op == "s_swappc_b64" for _, op in inside), sum(op in ("v_readlane_b32", "v_writelane_b32") for _, op in inside),
                sum(op.startswith("global_store") for _, op in inside)))
    want = int(os.environ.get("LOOP_SIZE", "0"))
    best = max(loops, key=size) if not want else min(loops, key=lambda lp: abs(size(lp) - want))
    body = [(a, op, ar) for a, op, ar in ins if best[0] <= a <= best[1]]
    return body


def hist(body):
    return collections.Counter(op for _, op, _ in body)


text = disasm(sys.argv[1])
h1 = hist(main_loop(text, sys.argv[2]))
print(sys.argv[2], "loop instructions:", sum(h1.values()))
if len(sys.argv) > 3:
    h2 = hist(main_loop(text, sys.argv[3]))
    print(sys.argv[3], "loop instructions:", sum(h2.values()))
    for op in sorted(set(h1) | set(h2), key=lambda o: -(h2[o] - h1[o])):
        if h2[op] != h1[op]:
            print("  %-28s %4d -> %4d  (%+d)" % (op, h1[op], h2[op], h2[op] - h1[op]))
else:
    for op, n in h1.most_common():
        print("  %-28s %4d" % (op, n))
