#!/usr/bin/env python3
"""profiles/<round>_pmc_k_accumulate[_381|_g2].json from the rocprofv3 passes of tools/profile_gpu.sh: the counters of the headline
k_accumulate_glds launch (the largest grid), what is derived from them, the identity of the kernel sources they were measured
on (bench.py quotes `roofline.traffic` only when that identity matches the library it runs) -- and, since round 3, the HBM-side
byte count CORRECTED with the calibration of tools/calib_fetch.hip (profiles/r03_calib_fetch.txt): on gfx950 FETCH_SIZE is
64 B x the number of read requests the L2 sends to the fabric, and a request is 64 OR 128 bytes (two adjacent sectors of a line
asked for together), so the raw figure under-counts by a shape-dependent factor:

    coalesced streams (16 or 8 B per lane)                     0.500
    quad-cooperative LDS-DMA gather, 128-B / 192-B / 256-B records   0.501 / 0.668 / 0.503
    one 64-B sector per lane (the entry queue's refill)        see calib_gather_lane<1>

  usage: make_pmc_json.py <prof dir> <out json> <curve: 377|381|g2> [calib table]
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha16  # noqa: E402

prof, out_json, curve = sys.argv[1], sys.argv[2], sys.argv[3]
calib_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "r03_calib_fetch.txt")   # (the TRACKED copy)
CFG = {
    "377": dict(npow=26, c=20, windows=13, record=192, law="twisted Edwards (7M)", gather="calib_gather_glds<3>", entry="calib_gather_lane<1>", entries_per_refill=8),
    "381": dict(npow=26, c=20, windows=13, record=128, law="XYZZ (8M + 2S)", gather="calib_gather_glds<2>", entry="calib_gather_lane<1>", entries_per_refill=4),
    # (round 5: two lanes per point, csrc/fp2pair.hpp -- the pair shares its entries, four per refill of the register queue)
    "g2": dict(npow=24, c=20, windows=13, record=256, law="XYZZ over Fq2, two lanes per point", gather="calib_gather_glds<4>", entry=None, entries_per_refill=4),
    "381g2": dict(npow=24, c=20, windows=13, record=256, law="XYZZ over Fq2 (BLS12-381), two lanes per point", gather="calib_gather_glds<4>", entry=None, entries_per_refill=4),
}[curve]
calib = {}
if os.path.exists(calib_path):
    for line in open(calib_path):
        f = line.split()
        if f and f[0].startswith("calib_") and len(f) >= 6:
            calib[f[0]] = {"fetch_ratio": float(f[3]), "write_ratio": float(f[5])}


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


def stage_counters():
    """Counters of the HBM-bound stage kernels of the same bench step (stage_roofline on counters, not on structural-byte
    arithmetic alone): per kernel name, every counter summed over the step's launches of that kernel, plus its time in the kernel trace."""
    want = {"k_l1_hist": "digits", "k_l1_scan": "digits", "k_l1_scatter": "digits", "k_pass_hist": "sort", "k_pass_scan": "sort", "k_pass_scatter": "sort",
            "k_bucket_reduce": "bucket_reduce"}
    per = {}
    for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES_SQ_INSTS_VALU_SQ_WAVE_CYCLES_SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU_SQ_WAIT_INST_ANY_SQ_WAIT_ANY_SQ_ACTIVE_INST_ANY",
                "GRBM_GUI_ACTIVE"):
        for f in glob.glob(os.path.join(prof, "pmc_" + tag, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r.get("Kernel_Name", "")
                for k in want:
                    if k in name:
                        d = per.setdefault(k, {})
                        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # time: the kernel-trace pass ran `steps` timed steps + warm-up: average per step = total / number of k_accumulate launches of the headline grid
    for f in glob.glob(os.path.join(prof, "stats", "**", "*kernel_trace.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        steps = sum(1 for r in rows if "k_accumulate" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == c.get("lanes")) or 1
        for r in rows:
            for k in want:
                if k in r["Kernel_Name"]:
                    per.setdefault(k, {})
                    per[k]["ms_per_step"] = per[k].get("ms_per_step", 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 / steps
    out = {}
    for k, d in per.items():
        row = {"stage": want[k], "ms_per_step": d.get("ms_per_step"), "FETCH_SIZE_KiB": d.get("FETCH_SIZE"), "WRITE_SIZE_KiB": d.get("WRITE_SIZE")}
        # coalesced streams: FETCH_SIZE tallies 64 B per request while a request carries 128 B (calibration: 0.500); the scattered 8-byte-entry
        # runs of the grouping are written as whole 64 / 128-B runs (WRITE_SIZE exact for coalesced stores)
        if d.get("FETCH_SIZE") is not None and d.get("WRITE_SIZE") is not None:
            row["hbm_bytes_estimate"] = d["FETCH_SIZE"] * 1024.0 / 0.5 + d["WRITE_SIZE"] * 1024.0
            if d.get("ms_per_step"):
                row["GBps"] = row["hbm_bytes_estimate"] / (d["ms_per_step"] * 1e-3) / 1e9
        if d.get("SQ_ACTIVE_INST_VALU") and d.get("GRBM_GUI_ACTIVE"):
            row["valu_busy_fraction"] = d["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (d["GRBM_GUI_ACTIVE"] / 8)
        out[k] = row
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
n = 1 << CFG["npow"]
# the plan of the profiled run (bench line of the kernel-trace pass): window size, windows and -- since the anchored window, whose top
# window stays (almost) empty -- the number of mixed additions of one launch as the device counted them
cfg_line = None
sb = os.path.join(prof, "stats_bench.json")
if os.path.exists(sb):
    lines = [ln for ln in open(sb).read().splitlines() if ln.startswith("{")]
    if lines:
        cfg_line = json.loads(lines[-1])["config"]
        CFG["c"], CFG["windows"] = cfg_line["window_bits"], cfg_line["windows"]
entries = CFG["windows"] * n * (1 - 2.0 ** -CFG["c"])     # non-zero digits
if cfg_line and cfg_line.get("mixed_additions_per_launch"):
    entries = float(cfg_line["mixed_additions_per_launch"])
fetch_raw = (c.get("FETCH_SIZE") or 0) * 1024.0
write_raw = (c.get("WRITE_SIZE") or 0) * 1024.0
# what the launch has to read, by shape: one base record per entry, the sorted entries (8 B each; a lane that takes one entry
# per load -- G2 -- asks for a whole sector each time unless the line survived in the L2)
base_bytes = entries * CFG["record"]
entry_bytes = entries * 8.0
r_gather = calib.get(CFG["gather"], {}).get("fetch_ratio")
r_entry = calib.get(CFG["entry"], {}).get("fetch_ratio") if CFG["entry"] else None
model = {"base_record_bytes": CFG["record"], "bases_bytes": base_bytes, "entries_bytes": entry_bytes,
         "algorithmic_bytes_per_pair_SURVEY_8d": 224 if curve in ("g2", "381g2") else 128,
         "structural_bytes_per_pair": (base_bytes + entry_bytes) / n,
         "calibration": {"file": os.path.relpath(calib_path, ROOT), "gather_ratio": r_gather, "entry_ratio": r_entry}}
corrected = None
if r_gather:
    if r_entry:
        predicted_raw = base_bytes * r_gather + entry_bytes * r_entry
        model["predicted_FETCH_SIZE_bytes_if_every_byte_is_read_once"] = predicted_raw
        model["measured_over_predicted"] = fetch_raw / predicted_raw if predicted_raw else None
        # scale the known composition by the measured excess: bytes actually requested from the fabric
        corrected = (base_bytes + entry_bytes) * (fetch_raw / predicted_raw) if predicted_raw else None
    else:
        # entries arrive one per load: every request beyond the base gathers is a 64-B sector fetched for 8 useful bytes
        sector_requests = max(0.0, fetch_raw - base_bytes * r_gather) / 64.0
        model["entry_sector_requests"] = sector_requests
        model["entry_sectors_per_entry"] = sector_requests / entries
        corrected = base_bytes + sector_requests * 64.0
# the execution plan the counters belong to (window size, entries per lane, lanes, group law live in msm_engine.hip's Plan, which the
# kernel-source hash does not cover -- ADVICE r3): taken from the bench line of the kernel-trace pass, compared by bench.py
plan = None
sb = os.path.join(prof, "stats_bench.json")
if os.path.exists(sb):
    lines = [ln for ln in open(sb).read().splitlines() if ln.startswith("{")]
    if lines:
        cfgj = json.loads(lines[-1])["config"]
        plan = {"window_bits": cfgj["window_bits"], "windows": cfgj["windows"], "lane_entries": cfgj["lane_entries"], "group_law": cfgj["group_law"],
                "lanes": c.get("lanes")}
res = {
    "plan": plan,
    "config": "bls12_%s npow=%d (c = %d, %d windows), the k_accumulate_glds launch of a bench step: %s, LDS-DMA quad-cooperative gathers of %d-B "
              "records, sorted (value, key) entries %s" % ({"377": "377_g1", "381": "381_g1", "g2": "377_g2", "381g2": "381_g2"}[curve], CFG["npow"], CFG["c"], CFG["windows"], CFG["law"],
                                                           CFG["record"], "through a register queue (%d per refill)" % CFG["entries_per_refill"] if CFG["entries_per_refill"] > 1 else "one per load"),
    "source": "tools/profile_gpu.sh: rocprofv3 --kernel-trace --pmc ..., one counter group per pass",
    "kernel_source_sha16": kernel_source_sha16(),
    "FETCH_SIZE_KiB": c.get("FETCH_SIZE"), "WRITE_SIZE_KiB": c.get("WRITE_SIZE"),
    "traffic_bytes_raw": fetch_raw + write_raw,
    "traffic_bytes_corrected": (corrected + write_raw) if corrected else None,
    "traffic_model": model,
    "note": "gfx950 FETCH_SIZE = 64 B x fabric read requests, a request being 64 or 128 B; `traffic_bytes_corrected` rescales the raw figure by the ratios "
            "measured on the same access shapes with known byte counts (tools/calib_fetch.hip). WRITE_SIZE is exact for coalesced stores and 1.28x for "
            "scattered 224-B records (partial sectors), left as reported. Infinity-Cache hits are counted, not excluded: this is traffic at the L2's "
            "fabric port, an upper bound on what HBM moves.",
    "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "SQ_WAVE_CYCLES_quad": c.get("SQ_WAVE_CYCLES"), "SQ_ACTIVE_INST_VALU_quad": c.get("SQ_ACTIVE_INST_VALU"),
    "SQ_WAIT_INST_ANY_quad": c.get("SQ_WAIT_INST_ANY"), "SQ_WAIT_ANY_quad": c.get("SQ_WAIT_ANY"), "GRBM_GUI_ACTIVE": c.get("GRBM_GUI_ACTIVE"),
    "kernel_ms_rocprof": kern_ms, "lanes": c.get("lanes"),
    "derived": {},
    "stages": stage_counters(),
    "stages_note": "per-kernel counters of the grouping and bucket-reduction kernels in the same passes (one bench step): FETCH_SIZE / WRITE_SIZE as "
                   "reported (KiB), hbm_bytes_estimate = FETCH_SIZE / 0.5 + WRITE_SIZE (these kernels read coalesced streams, which gfx950 counts at "
                   "64 B per 128-B request: tools/calib_fetch.hip); bench.py's stage_roofline quotes them next to its structural-byte figures",
}
if kern_ms and c.get("GRBM_GUI_ACTIVE"):
    res["derived"]["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (kern_ms * 1e-3) / 1e9
if c.get("SQ_INSTS_VALU"):
    res["derived"]["valu_instr_per_mixed_add"] = c["SQ_INSTS_VALU"] * 64 / entries
if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE"):
    res["derived"]["valu_busy_fraction"] = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8)
if kern_ms and corrected:
    res["derived"]["fabric_read_TBps"] = corrected / (kern_ms * 1e-3) / 1e12
json.dump(res, open(out_json, "w"), indent=1)
print(json.dumps(res, indent=1))
