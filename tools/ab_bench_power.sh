#!/bin/bash
# tools/ab_bench.sh with socket power / shader clock sampled during every run (tools/power_probe.sh).
# usage: tools/ab_bench_power.sh [rounds] [bench args...]
cd "$(dirname "$0")/.."
ROUNDS=${1:-2}; shift
LIB=2022-entries_amd/libmi355msm.so
cp $LIB /tmp/keep.so
for r in $(seq $ROUNDS); do
  for v in 2022-entries_amd/build/variants/*.so; do
    cp $v $LIB
    echo -n "$(basename $v .so) r$r: "
    bash tools/power_probe.sh /tmp/pw.txt -- timeout 300 python bench.py --steps 8 --warmup 2 --cpu-sample-pow 0 --extras 0 --also-precompute 0 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms_per_step']
print('step %.2f ms  accumulate %.2f  sort %.2f  reduce %.2f' % (j['ms_per_step'], s['accumulate'], s['sort'], s['bucket_reduce']), end='')"
    python - <<'PY'
import re
vals = []
for line in open("/tmp/pw.txt"):
    m = re.search(r"sclk.*?\((\d+)Mhz\)", line); p = re.search(r"Power.*?:\s*([0-9.]+)", line)
    if m and p: vals.append((int(m.group(1)), float(p.group(1))))
busy = [v for v in vals if v[1] > 600]
if busy:
    print("   busy (%d samples): sclk %.0f MHz, power %.0f W" % (len(busy), sum(v[0] for v in busy) / len(busy), sum(v[1] for v in busy) / len(busy)))
else:
    print()
PY
  done
done
cp /tmp/keep.so $LIB
