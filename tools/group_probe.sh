#!/bin/bash
# Run on the GPU box: same-box timing of grouping variants (tools/gp/pt_*, builds of tools/partition_test.hip with different -D
# switches) at 2^26 pairs, c = 20, interleaved over ROUNDS rounds; then the correctness matrix of every variant named in CHECK.
#   tools/group_probe.sh "base digits fused" "fused"
cd "$(dirname "$0")/.."
VARIANTS=${1:-base}
CHECK=${2:-}
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
  for v in $VARIANTS; do
    echo "== $v (round $r)"
    timeout 120 tools/gp/pt_$v 26 20 0 0 4 | grep "^grouping\|^per kernel"
  done
done
for v in $CHECK; do
  echo "== correctness $v"
  for args in "10 9 0 0" "16 13 0 0" "20 14 0 0" "20 14 0 1" "20 14 1 0" "22 16 0 0" "22 20 0 2" "-70001 11 0 0" "21 15 0 3" "22 17 1 1" "18 8 0 0"; do
    echo -n "  [$args] "; timeout 300 tools/gp/pt_$v $args 1 | grep "RESULT\|FAIL" | head -3 | tr '\n' ' '; echo
  done
done
