#!/bin/bash
# On the GPU box: the headline workload against the window size (no tables), with power / clock sampled during each run.
# profiles/r03_ab_window.txt
cd "$(dirname "$0")/.."
for C in 19 20 21 22 23; do
  echo -n "window_bits $C: "
  bash tools/power_probe.sh /tmp/pw_$C.txt -- python bench.py --steps 6 --warmup 2 --cpu-sample-pow 0 --extras 0 --also-precompute 0 --window-bits $C 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j['stage_ms_per_step']
print('step %.2f ms  accumulate %.2f  sort %.2f  digits %.2f  merge %.2f  reduce %.2f  windows %d' % (j['ms_per_step'], s['accumulate'], s['sort'], s['digits'], s['segreduce'], s['bucket_reduce'], j['config']['windows']))"
  python - <<PY
import re
vals = []
for line in open("/tmp/pw_$C.txt"):
    m = re.search(r"sclk.*?\((\d+)Mhz\)", line); p = re.search(r"Power.*?:\s*([0-9.]+)", line)
    if m and p: vals.append((int(m.group(1)), float(p.group(1))))
busy = [v for v in vals if v[1] > 600]
if busy:
    print("    while busy (%d samples): sclk %.0f MHz avg, power %.0f W avg" % (len(busy), sum(v[0] for v in busy) / len(busy), sum(v[1] for v in busy) / len(busy)))
PY
done
