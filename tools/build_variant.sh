#!/bin/bash
# Experimental build of the library: tools/build_variant.sh NAME "FLAGS" "TU ..." compiles the listed translation units
# (e.g. "aslr_forward_nj2 aslr_backward_nx8") with the extra FLAGS and links them with the product objects of the others
# into tools/ubench/libaslr_to_hip_NAME.so (selected with ASLR_LIB_OVERRIDE by tools/time_*.py; never loaded by the package).
set -e
NAME=$1; FLAGS=$2; TUS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/aslr_to_amd/csrc
make -s -j8 -C $C
O=/tmp/aslr_variant_$NAME; mkdir -p $O
ALL="aslr_abi aslr_calc_nj2 aslr_calc_nj7 aslr_calc_nj7_vsa aslr_backward_nx8 aslr_backward_nx28 aslr_forward_nj2 aslr_forward_nj7"
OBJS=""
for tu in $ALL; do
  if echo " $TUS " | grep -q " $tu "; then
    (cd $C && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed $FLAGS -c $tu.hip -o $O/$tu.o) &
    OBJS="$OBJS $O/$tu.o"
  else
    OBJS="$OBJS $C/$tu.o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -o $R/tools/ubench/libaslr_to_hip_$NAME.so $OBJS
echo built tools/ubench/libaslr_to_hip_$NAME.so
