#!/bin/bash
# On the GPU box: entries per accumulate lane (option lane_entries) at the headline size -- the tail of the launch (blocks of
# 256 lanes, 768 resident) against the fragment-merge cost.  Interleaved twice.
cd "$(dirname "$0")/.."
for r in 1 2; do
for K in 256 128 160 192 224 240 264 288 320 384 512; do
  echo -n "lane_entries $K r$r: "
  python bench.py --steps 6 --warmup 2 --cpu-sample-pow 0 --extras 0 --also-precompute 0 --lane-entries $K 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms_per_step']
print('step %.2f ms  accumulate %.2f  merge %.2f  reduce %.2f  sort %.2f' % (j['ms_per_step'], s['accumulate'], s['segreduce'], s['bucket_reduce'], s['sort']))"
done
done
