#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of three 2^26 host-scalar batches (profiles/r03_rocprof_host_scalars.txt).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/prof_host
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_host -- python tools/host_batch_once.py > gpurun_out/prof_host.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_host/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = ["# rocprofv3 --kernel-trace --stats -- python tools/host_batch_once.py: three 2^26 host-scalar batches (BLS12-377 G1), each as 3 carried chunks (1/13 + 3/13 + 9/13)",
       "%-70s %6s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "%")]
for r in rows[:22]:
    out.append("%-70s %6s %12.3f %12.4f %7s" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
open("gpurun_out/r03_rocprof_host_scalars.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
find gpurun_out/prof_host -name "*.csv" -size +1M -delete
