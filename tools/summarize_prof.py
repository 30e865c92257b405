#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into the per-kernel summary committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    for k in ("k_accumulate", "k_segreduce", "k_bucket_reduce", "k_l1_hist", "k_l1_scatter", "k_l1_scan", "k_l1_merge", "k_pass_hist", "k_pass_scan",
              "k_pass_scatter", "k_pass_subjobs", "k_digits", "k_convert_bases"):
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
print("== k_accumulate per launch configuration (kernel trace of the same run; grid = lanes) ==")
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True):
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_accumulate" in r["Kernel_Name"]:
            per[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(r["VGPR_Count"]), int(r["Scratch_Size"]))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for (k, grid, vg, scr), v in sorted(per.items()):
        print("%-22s lanes=%-9d scratch=%-3d launches=%-3d avg_ms=%.3f min=%.3f max=%.3f" % (k, grid, scr, len(v), sum(v) / len(v), min(v), max(v)))

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
        rows = list(csv.DictReader(open(f)))
        perd = defaultdict(lambda: defaultdict(float))
        grid = {}
        for r in rows:
            if "k_accumulate" in r.get("Kernel_Name", ""):
                perd[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
                grid[r["Dispatch_Id"]] = r.get("Grid_Size", "?")
        for did in sorted(perd, key=int):
            print("   k_accumulate dispatch %-5s lanes=%-9s %s" % (did, grid[did], "  ".join("%s=%.6g" % kv for kv in sorted(perd[did].items()))))
        for k in sorted(agg, key=lambda k: -max(agg[k].values())):
            print("   %-56s dispatches=%-4d %s" % (k, len(disp[k]), "  ".join("%s=%.6g" % kv for kv in sorted(agg[k].items()))))
