#!/bin/bash
# developer tool: build libfcd variants with -DFCD_EXP=n (timing experiments; results may be wrong)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Iinclude"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DFCD_EXP=$n -c fast_ctc_decode_amd/csrc/beam_wave.hip -o build_exp/beam_wave_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_exp/libfcd_exp_$n.so build_exp/beam_wave_$n.o \
     fast_ctc_decode_amd/csrc/capi.o fast_ctc_decode_amd/csrc/beam_generic.o fast_ctc_decode_amd/csrc/viterbi.o fast_ctc_decode_amd/csrc/duplex.o
done
