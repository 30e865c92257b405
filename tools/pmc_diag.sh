#!/bin/bash
# On the GPU box: diagnostic counter passes for the accumulate kernel of one curve (one counter group per pass, kernel-trace only).
#   EXTRA="--curve bls12_377_g2" NPOW=24 bash tools/pmc_diag.sh "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_IFETCH_LEVEL" ...
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/pmc_diag}
rm -rf $OUT; mkdir -p $OUT
NPOW=${NPOW:-26}
EXTRA=${EXTRA:-}
for C in "$@"; do
  TAG=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$TAG -- python bench.py --steps 1 --warmup 0 --cpu-sample-pow 0 --extras 0 --also-precompute 0 --npow $NPOW $EXTRA > /dev/null 2> $OUT/$TAG.err
  python - "$OUT/$TAG" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float)
    n = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_accumulate" not in k: continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    for c, v in acc.items():
        print("%-32s %18.0f  (sum over %d dispatches)" % (c, v, n[c]))
PY
done
