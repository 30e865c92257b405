#!/bin/bash
# On the GPU box: the assume_subgroup option (scalars above r/2 folded to r - k) against the window size, on the engine variants
# staged under 2022-entries_amd/build/variants/*.so (a_base: level 1 resolves 9 bucket bits; hb10: 10, so that c = 21 needs one
# generic pass instead of two).  Same box, interleaved.  profiles/r03_ab_fold.txt
cd "$(dirname "$0")/.."
LIB=2022-entries_amd/libmi355msm.so
cp $LIB /tmp/keep.so
run() {
  timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample-pow 0 --extras 0 --also-precompute 0 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms_per_step']
print('step %.2f ms  accumulate %.2f  sort %.2f  digits %.2f  merge %.2f  reduce %.2f  c %d windows %d' % (j['ms_per_step'], s['accumulate'], s['sort'], s['digits'], s['segreduce'], s['bucket_reduce'], j['config']['window_bits'], j['config']['windows']))"
}
for r in 1 2; do
  for v in 2022-entries_amd/build/variants/*.so; do
    cp $v $LIB
    for args in "--window-bits 20" "--window-bits 21" "--window-bits 21 --assume-subgroup 1" "--window-bits 22 --assume-subgroup 1" "--assume-subgroup 1" "$@"; do
      [ -z "$args" ] && continue
      echo -n "$(basename $v .so) r$r [$args]: "
      run $args
    done
  done
done
cp /tmp/keep.so $LIB
