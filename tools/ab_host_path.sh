#!/bin/bash
# On the GPU box: the host-scalar path (SURVEY 8d's primary metric) under two builds of the engine, interleaved: the library named by $1
# (default: libmi355msm_merge_variant.so = carried batches with a merge pass per piece, rounds 3-5) against the tree's own.
# Each line: tools/host_stage_probe.py (wall ms and stage sums for scalars on the device, in pageable and in pinned host memory).
cd "$(dirname "$0")/.."
V=${1:-libmi355msm_merge_variant.so}
for r in 1 2 3; do
  echo "== round $r: variant $V"; MI355_MSM_LIBRARY=$V python tools/host_stage_probe.py 2>&1 | grep div
  echo "== round $r: this tree";  python tools/host_stage_probe.py 2>&1 | grep div
done
