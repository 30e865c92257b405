#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into the per-kernel summary committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    for k in ("k_accumulate", "k_segreduce", "k_bucket_reduce", "k_digits", "k_convert_bases"):
        if k in name:
            return k + ("<381>" if "381" in name else "<377>" if "377" in name else "")
    if "rocprim" in name or "radix" in name or "onesweep" in name:
        return "rocprim:" + name.split("(")[0].split("::")[-1][:48]
    return name[:60]


print("== kernel stats (rocprofv3 --kernel-trace --stats; bench.py --steps 3 --warmup 1) ==")
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        k = short(r["Name"])
        agg[k][0] += int(r["Calls"])
        agg[k][1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    print("%-60s %8s %14s %14s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "%"))
    for k, (calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-60s %8d %14.3f %14.4f %6.1f%%" % (k, calls, ns / 1e6, ns / 1e6 / calls, 100 * ns / tot))

print()
print("== PMC passes (per-kernel sums over one bench step; FETCH_SIZE/WRITE_SIZE in KiB as reported) ==")
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", r.get("Name", "?")))
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r.get("Dispatch_Id", "0"))
        print("-- %s" % os.path.basename(d))
        for k in sorted(agg, key=lambda k: -max(agg[k].values())):
            print("   %-56s dispatches=%-4d %s" % (k, len(disp[k]), "  ".join("%s=%.6g" % kv for kv in sorted(agg[k].items()))))
