#!/bin/bash
# On the GPU box: entries per accumulate lane on BLS12-381 2^26, BLS12-377 2^25 and 2^26 (profiles/r03_ab_lane_entries.txt).
cd "$(dirname "$0")/.."
run() { python bench.py --steps 6 --warmup 2 --cpu-sample-pow 0 --extras 0 --also-precompute 0 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms_per_step']
print('step %.2f ms  accumulate %.2f  merge %.2f  reduce %.2f' % (j['ms_per_step'], s['accumulate'], s['segreduce'], s['bucket_reduce']))"; }
for r in 1 2; do
for K in 256 512 384; do echo -n "381 2^26 K=$K r$r: "; run --curve bls12_381_g1 --lane-entries $K; done
for K in 256 416 512; do echo -n "377 2^25 K=$K r$r: "; run --npow 25 --lane-entries $K; done
for K in 256 512 1024; do echo -n "377 2^26 K=$K r$r: "; run --lane-entries $K; done
done
