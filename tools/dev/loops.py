"""tools/dev/loops.py OBJECT.o KERNEL-SUBSTRING [ADDR]: every backward-branch loop of a kernel with its instruction mix (f64, LDS,
v_readlane / v_writelane = scalar-register spills, DPP, waits, calls); with ADDR (hex, a loop's first address) the
loop's instructions.  Developer aid for the lone-wavefront kernels, whose cost is their instruction count."""
import re, subprocess, sys, tempfile
B = "/opt/rocm/lib/llvm/bin"
def disasm(obj):
    t = tempfile.mkdtemp()
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, t + "/f.bin"])
    subprocess.check_call([B + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + t + "/f.bin", "--output=" + t + "/d.co", "--unbundle"])
    return subprocess.check_output([B + "/llvm-objdump", "-d", t + "/d.co"]).decode()
text = disasm(sys.argv[1])
m = [x for x in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^\n|\Z)", text, re.S | re.M) if sys.argv[2] in x.group(1)][0]
ins = []
for l in m.group(2).splitlines():
    mm = re.match(r"\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):", l)
    if mm:
        ins.append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
loops = set()
for a, op, args in ins:
    if op.startswith("s_cbranch") or op == "s_branch":
        off = int(args.split()[-1])
        if off >= 32768:
            loops.add((a + 4 + (off - 65536) * 4, a))
want = int(sys.argv[3], 16) if len(sys.argv) > 3 else None
for lp in sorted(loops):
    inside = [(a, op, ar) for a, op, ar in ins if lp[0] <= a <= lp[1]]
    if want is None:
        c = lambda f: sum(1 for _, op, _ in inside if f(op))
        print("loop %6x..%6x %5d instr, f64 %3d, ds_read %2d ds_write %2d, readlane %3d writelane %3d, dpp %d, waitcnt %d, calls %d, vmem %d" % (
            lp[0], lp[1], len(inside), c(lambda o: "_f64" in o), c(lambda o: o.startswith("ds_read")), c(lambda o: o.startswith("ds_write")),
            c(lambda o: o == "v_readlane_b32"), c(lambda o: o == "v_writelane_b32"), c(lambda o: "dpp" in o), c(lambda o: o == "s_waitcnt"),
            c(lambda o: o == "s_swappc_b64"), c(lambda o: o.startswith(("global_", "flat_", "buffer_")))))
    elif lp[0] == want:
        for a, op, ar in inside:
            print("  %6x  %s %s" % (a, op, ar))
        break
