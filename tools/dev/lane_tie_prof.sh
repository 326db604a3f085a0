#!/bin/bash
# developer aid: builds tools/dev/_build/libfcd_hip_prof.so = the library with beam_lane.hip compiled -DFCD_LANE_TIE_PROF
# (shader-clock stamps inside the tie-flagged step of the wide-beam kernel; read back by tools/dev/lane_tie_prof.py)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/fast_ctc_decode_amd/csrc
mkdir -p $R/tools/dev/_build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -DFCD_LANE_TIE_PROF=1 -c $C/beam_lane.hip -o $R/tools/dev/_build/beam_lane_prof.o
OBJS=$(ls $C/*.o | grep -v beam_lane.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/dev/_build/libfcd_hip_prof.so $OBJS $R/tools/dev/_build/beam_lane_prof.o
echo $R/tools/dev/_build/libfcd_hip_prof.so
