#!/bin/bash
# Stage an engine variant for tools/ab_bench.sh / tools/ab_variants.sh: recompile kernel units with extra -D flags and link them with
# the other objects of the current build into 2022-entries_amd/build/variants/<name>.so.
#   tools/build_variant.sh noq -DTE_ENTRY_Q=0 -DMSM_SW_ENTRY_Q=0
#   UNITS="kernels_377g2p" tools/build_variant.sh g2p_eq4 -DMSM_G2P_ENTRY_Q=4      (only these units are recompiled)
set -e
cd "$(dirname "$0")/../2022-entries_amd"
NAME=$1; shift
UNITS=${UNITS:-kernels_377te kernels_377g1 kernels_381g1 kernels_377g2}
mkdir -p build/variants build/var_$NAME
OBJS=""
for u in msm_engine partition kernels_377g1 kernels_381g1 kernels_377g2 kernels_381g2 kernels_377te kernels_377g2p kernels_381g2p; do
  if echo " $UNITS " | grep -q " $u "; then
    hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC "$@" -c csrc/$u.hip -o build/var_$NAME/$u.o &
    OBJS="$OBJS build/var_$NAME/$u.o"
  else
    OBJS="$OBJS build/$u.o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$NAME.so $OBJS
ls -la build/variants/$NAME.so
