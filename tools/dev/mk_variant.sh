#!/bin/bash
# tools/dev/mk_variant.sh NAME [SOURCE.hip ...]: recompiles the named translation units (default beam_wave.hip) of the working
# tree and links them with the other objects of the last full build into ab_variants/NAME.so -- an A/B candidate that
# tools/dev/time_variant.py / time_duplex.py load by path on the GPU box (developer scratch; ab_variants/ is not committed).
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
SRCS=${@:-beam_wave.hip}
C=fast_ctc_decode_amd/csrc
mkdir -p ab_variants/obj_$NAME
FLAGS="$EXTRA --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function"
OBJS=""
for o in $C/*.o; do
  b=$(basename $o .o)
  if echo " $SRCS " | grep -q " $b.hip "; then
    /opt/rocm/bin/hipcc $FLAGS -c $C/$b.hip -o ab_variants/obj_$NAME/$b.o &
    OBJS="$OBJS ab_variants/obj_$NAME/$b.o"
  else
    OBJS="$OBJS $o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_variants/$NAME.so $OBJS
echo ab_variants/$NAME.so
