#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE per calibration kernel of tools/calib_fetch.hip against the bytes each one is known to move.

  python tools/calib_fetch_summary.py <dir with the rocprofv3 passes and calib_expected.txt>
"""
import csv
import glob
import os
import sys

d = sys.argv[1]
expected = {}
for line in open(os.path.join(d, "calib_expected.txt")):
    if line.startswith("calib_"):
        k, v = line.split()
        expected[k] = int(v)
got = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        for k in expected:
            base = k.split("<")[0]
            if base in name and (("<" not in k) or (k.split("<")[1].rstrip(">") + ">" in name.replace(" ", "")) or ("<" + k.split("<")[1]) in name.replace(" ", "")):
                got.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
                got[k][r["Counter_Name"]] += float(r["Counter_Value"])
print("# FETCH_SIZE / WRITE_SIZE are reported in KiB; ratio = counter bytes / bytes the kernel is known to move")
print("%-24s %16s %16s %8s %16s %8s" % ("kernel", "known bytes", "FETCH_SIZE B", "ratio", "WRITE_SIZE B", "ratio"))
for k, e in expected.items():
    c = got.get(k, {})
    f = c.get("FETCH_SIZE", float("nan")) * 1024
    w = c.get("WRITE_SIZE", float("nan")) * 1024
    print("%-24s %16d %16.0f %8.3f %16.0f %8.3f" % (k, e, f, f / e, w, w / e))
