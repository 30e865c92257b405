#!/bin/bash
# Stage an engine variant for tools/ab_bench.sh: recompile the kernel units with extra -D flags and link them with the
# other objects of the current build into 2022-entries_amd/build/variants/<name>.so.
#   tools/build_variant.sh noq -DTE_ENTRY_Q=0 -DMSM_SW_ENTRY_Q=0
set -e
cd "$(dirname "$0")/../2022-entries_amd"
NAME=$1; shift
mkdir -p build/variants build/var_$NAME
for u in kernels_377te kernels_377g1 kernels_381g1 kernels_377g2; do
  hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC "$@" -c csrc/$u.hip -o build/var_$NAME/$u.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$NAME.so build/msm_engine.o build/partition.o build/var_$NAME/*.o
ls -la build/variants/$NAME.so
