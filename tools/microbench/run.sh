#!/bin/bash
# Build (here or on the GPU box) and run the micro-benchmarks behind DESIGN.md's "what bounds it" paragraphs:
#   readbw    what a pure streaming read reaches on this GPU (context for the viterbi kernel's HBM fraction)
#   vitshape  the viterbi kernel's memory shape -- tile loads, output stores -- without its arithmetic
#   rowloop   the duplex max-mode row loop: cycles per row with / without LDS operands, stores, idle lanes
# Usage: tools/microbench/run.sh [readbw|vitshape|rowloop ...]      (default: all three)
set -e
cd "$(dirname "$0")"
for b in ${@:-readbw vitshape rowloop}; do
    [ -x $b ] && [ $b -nt $b.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $b.hip -o $b
    echo "== $b"
    ./$b
done
