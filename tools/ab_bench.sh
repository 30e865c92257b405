#!/bin/bash
# Run on the GPU box: A/B the engine variants staged under 2022-entries_amd/build/variants/*.so on the SAME box,
# interleaved, so that box-to-box and DVFS noise cancels.  usage: tools/ab_bench.sh [rounds] [bench args...]
cd "$(dirname "$0")/.."
ROUNDS=${1:-2}; shift
LIB=2022-entries_amd/libmi355msm.so
cp $LIB /tmp/keep.so
for r in $(seq $ROUNDS); do
  for v in 2022-entries_amd/build/variants/*.so; do
    cp $v $LIB
    echo -n "$(basename $v .so) r$r: "
    timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample-pow 0 --extras 0 --also-precompute 0 "$@" | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms_per_step']
print('step %.2f ms  accumulate %.2f  sort %.2f  reduce %.2f' % (j['ms_per_step'], s['accumulate'], s['sort'], s['bucket_reduce']))"
  done
done
cp /tmp/keep.so $LIB
