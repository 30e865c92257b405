#!/usr/bin/env python3
"""profiles/r02_pmc_k_accumulate.json from the rocprofv3 passes of tools/profile_gpu.sh (gpurun_out/prof): the counters of the
headline k_accumulate_glds launch (the largest grid), what is derived from them, and the identity of the kernel sources they
were measured on (bench.py quotes `roofline.traffic` only when that identity matches the library it runs)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha16  # noqa: E402

prof = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof")
# optional: <out json> <windows> <npow> <window bits> <config text> -- for a second workload (G2: profiles/r02_pmc_k_accumulate_g2.json)
out_json = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r02_pmc_k_accumulate.json")
W_, NPOW_, C_ = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (13, 26, 20)
config_text = sys.argv[6] if len(sys.argv) > 6 else None


def headline_counters(tag):
    out = {}
    for f in glob.glob(os.path.join(prof, "pmc_" + tag, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if "k_accumulate" in r.get("Kernel_Name", ""):
                d = per.setdefault(r["Dispatch_Id"], {"grid": int(r.get("Grid_Size", 0))})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if per:
            best = max(per.values(), key=lambda d: d["grid"])
            out.update({k: v for k, v in best.items() if k != "grid"})
            out["lanes"] = best["grid"]
    return out


c = {}
for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES_SQ_INSTS_VALU_SQ_WAVE_CYCLES_SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU_SQ_WAIT_INST_ANY_SQ_WAIT_ANY_SQ_ACTIVE_INST_ANY",
            "GRBM_GUI_ACTIVE"):
    c.update(headline_counters(tag))
kern_ms = None
for f in glob.glob(os.path.join(prof, "stats", "**", "*kernel_trace.csv"), recursive=True):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f))
         if "k_accumulate" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == c.get("lanes")]
    if d:
        kern_ms = sum(d) / len(d)
entries = W_ * (1 << NPOW_) * (1 - 2.0 ** -C_)     # non-zero digits (c = 20, 13 windows at 2^26)
res = {
    "config": config_text or "bls12_377_g1 npow=26 (c = 20, 13 windows), the k_accumulate_glds<TeLaw> launch of a bench step: twisted-Edwards image, "
              "LDS-DMA quad-cooperative gathers of 192-B records, (value, key) entry stream",
    "source": "tools/profile_gpu.sh: rocprofv3 --kernel-trace --pmc ..., one counter group per pass; summary in profiles/r02_rocprof_summary.txt",
    "kernel_source_sha16": kernel_source_sha16(),
    "FETCH_SIZE_KiB": c.get("FETCH_SIZE"), "WRITE_SIZE_KiB": c.get("WRITE_SIZE"),
    "traffic_bytes_raw": (c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024.0,
    "note": "gfx950 FETCH_SIZE counts one 64-B unit per EA read request (coalesced 16 B/lane streams read 0.5x, see MI355X_MICROARCH.md); the "
            "cooperative gather requests whole 64-B sectors (three per record) plus the 8-B entries, so the raw figure is taken as the byte count",
    "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "SQ_WAVE_CYCLES_quad": c.get("SQ_WAVE_CYCLES"), "SQ_ACTIVE_INST_VALU_quad": c.get("SQ_ACTIVE_INST_VALU"),
    "SQ_WAIT_INST_ANY_quad": c.get("SQ_WAIT_INST_ANY"), "SQ_WAIT_ANY_quad": c.get("SQ_WAIT_ANY"), "GRBM_GUI_ACTIVE": c.get("GRBM_GUI_ACTIVE"),
    "kernel_ms_rocprof": kern_ms, "lanes": c.get("lanes"),
    "derived": {},
}
if kern_ms and c.get("GRBM_GUI_ACTIVE"):
    res["derived"]["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (kern_ms * 1e-3) / 1e9
if c.get("SQ_INSTS_VALU"):
    res["derived"]["valu_instr_per_mixed_add"] = c["SQ_INSTS_VALU"] * 64 / entries
if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE"):
    res["derived"]["valu_busy_fraction"] = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8)
json.dump(res, open(out_json, "w"), indent=1)
print(json.dumps(res, indent=1))
