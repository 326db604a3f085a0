#!/bin/bash
# tools/isa_table.sh > profiles/TAG_isa_table.txt : registers, scratch, LDS and instruction counts of the search kernels'
# instantiations (hipcc's code objects under fast_ctc_decode_amd/csrc/*.o; no GPU needed), with the wavefronts per
# SIMD their registers allow (512 VGPRs per SIMD lane, granules of 8, at most 8 wavefronts).
cd "$(dirname "$0")/.."
echo "# kernel instantiation | VGPR | SGPR | scratch bytes | LDS bytes per workgroup | instructions | wavefronts per SIMD by VGPR"
echo "# beam_wave_kernel<N, GW, RPW, S, AMB, PROF, UNI, H16, PDQ>; beam_lane_kernel<N, RPW, AMB, CRF, PDQ>; duplex_slots_kernel<MODE, PROF>; duplex_kernel<MODE, PIN>"
for f in beam_wave beam_lane beam_generic duplex_slots duplex viterbi; do
  tools/isa.sh fast_ctc_decode_amd/csrc/$f.o | python3 -c '
import re, subprocess, sys
for line in sys.stdin:
    name, rest = line.split(" vgpr", 1)
    name = name.strip()
    m = re.search(r"beam_wave_kernelI((?:L[ib]n?\d+E)+)", name)
    if m:
        args = [("-" if a.startswith("n") else "") + a.lstrip("n") for a in re.findall(r"L[ib](n?\d+)E", m.group(1))]
        args = [a if i < 4 else ("true" if a == "1" else "false") for i, a in enumerate(args)]
        name = "beam_wave_kernel<" + ", ".join(args) + ">"
    else:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"fcd::\(anonymous namespace\)::", "", name)
        name = re.sub(r"^void ", "", re.sub(r"\(.*$", "", name))
    v = int(rest.split()[0])
    g = max(8, (v + 7) // 8 * 8)
    print("%-64s vgpr%s  waves/SIMD %d" % (name, rest.rstrip(), min(8, 512 // g)))
'
done
