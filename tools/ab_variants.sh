#!/bin/bash
# Run on the GPU box: the same command under every engine variant staged in 2022-entries_amd/build/variants/*.so (and the tree's own
# library as "tree"), interleaved over `rounds`.   usage: tools/ab_variants.sh <rounds> <command...>
cd "$(dirname "$0")/.."
ROUNDS=${1:-2}; shift
LIB=2022-entries_amd/libmi355msm.so
cp $LIB /tmp/keep.so
for r in $(seq $ROUNDS); do
  for v in /tmp/keep.so 2022-entries_amd/build/variants/*.so; do
    cp $v $LIB
    n=$(basename $v .so); [ "$n" = keep ] && n=tree
    echo "== $n r$r"
    timeout 600 "$@"
  done
done
cp /tmp/keep.so $LIB
