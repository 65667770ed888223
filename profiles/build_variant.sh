#!/bin/bash
# Build an A/B variant of the library: bash profiles/build_variant.sh NAME UNIT "EXTRA FLAGS"
#   -> emd_amd/csrc/variants/lib_NAME.so = the in-tree objects with UNIT.hip recompiled with the extra -D flags (run `make -C emd_amd/csrc` first).
# Selected at run time with EMD_LIB_PATH (profiles/ab_variants.sh).
set -e
NAME=$1; UNIT=$2; EXTRA=$3
C=$(dirname "$0")/../emd_amd/csrc
mkdir -p $C/variants
COMMON="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -I$C/../../include"
case $UNIT in
  preprocess) UF="-ffp-contract=off -fno-slp-vectorize";;
  render|hexplane) UF="-fno-slp-vectorize";;
  *) UF="";;
esac
/opt/rocm/bin/hipcc $COMMON $UF $EXTRA -c $C/$UNIT.hip -o $C/variants/${UNIT}_$NAME.o 2> $C/variants/${UNIT}_$NAME.log || { tail -20 $C/variants/${UNIT}_$NAME.log; exit 1; }
OBJS=""
for o in api preprocess binning render sky loss hexplane embed optim densify mlp exchange; do
  if [ $o = $UNIT ]; then OBJS="$OBJS $C/variants/${UNIT}_$NAME.o"; else OBJS="$OBJS $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/lib_$NAME.so $OBJS
echo built $C/variants/lib_$NAME.so
