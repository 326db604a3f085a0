#!/bin/bash
# tools/isa.sh OBJECT.o [KERNEL-NAME-SUBSTRING]
# Pulls the gfx950 code object out of a hipcc object file and prints, per kernel whose (mangled) name contains the
# substring: VGPRs, SGPRs, scratch bytes, LDS bytes and the instruction count of its disassembly.  With
# ISA_DUMP=dir the disassembly of each matching kernel is written to dir/<name>.s (for diffing two builds).
set -e
B=/opt/rocm/lib/llvm/bin
OBJ=$1
PAT=${2:-}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
objcopy -O binary --only-section=.hip_fatbin "$OBJ" "$TMP/fat.bin"
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$TMP/fat.bin" --output="$TMP/dev.co" --unbundle
$B/llvm-readelf --notes "$TMP/dev.co" > "$TMP/notes.txt"
$B/llvm-objdump -d --no-show-raw-insn "$TMP/dev.co" > "$TMP/dis.s"
python3 - "$TMP/notes.txt" "$TMP/dis.s" "$PAT" "${ISA_DUMP:-}" <<'EOF'
import os, re, sys
notes, dis, pat, dump = open(sys.argv[1]).read(), open(sys.argv[2]).read(), sys.argv[3], sys.argv[4]
meta = {}
for blk in re.split(r"\n\s*- \.agpr_count", notes)[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name:
        continue
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
    meta[name.group(1)] = (g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"))
funcs = {}
for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^\n|\Z)", dis, re.S | re.M):
    funcs[m.group(1)] = m.group(2)
for name in sorted(meta):
    if pat and pat not in name:
        continue
    body = funcs.get(name, "")
    n = sum(1 for l in body.splitlines() if re.match(r"\s+[a-z_]", l))
    v, s, scr, lds = meta[name]
    print("%-90s vgpr %3d sgpr %3d scratch %4d lds %5d insts %5d" % (name[:90], v, s, scr, lds, n))
    if dump:
        os.makedirs(dump, exist_ok=True)
        open(os.path.join(dump, name[:120] + ".s"), "w").write(re.sub(r"^\s*//.*$", "", body, flags=re.M))
EOF
