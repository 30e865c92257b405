#!/usr/bin/env python3
"""On the GPU box: re-derive the constants of the window-size model (csrc/msm_engine.hip choose_window_bits) from measurements.

The model prices a plan in field-multiplication units: 10 per mixed addition (entry) + BUCKET per bucket of the bucket -> window
reduction (+ GROUP per entry and window from 22 bits on, where the grouping needs a second generic pass).  This tool measures, per
curve, the stage times of one MSM at the size given for every window size around the automatic one and prints
  * ns per entry of the accumulation and ns per bucket of merge + reduction  ->  BUCKET = 10 * (ns per bucket) / (ns per entry),
  * the extra grouping time per entry at c >= 22                             ->  GROUP,
  * for each size of a sweep, the measured-fastest c next to the model's choice (what tests/test_gpu_window_model.py guards),
and a C++ snippet with the constants to paste into choose_window_bits.
usage: tools/calibrate_window_model.py [curve=bls12_377_g1] [npows=20,22,24,26]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entries_amd as ea
import bench

curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_377_g1"
npows = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "20,22,24,26").split(",")]
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
top = bench.R381_TOP if "381" in curve else bench.R377_TOP
bucket_units, group_units = [], []
for npow in npows:
    n = 1 << npow
    ctx = ea.MultiScalarMultContext(curve)
    ctx.set_bases(tile.repeat(n >> 15, 1).contiguous())
    sc = bench.uniform_scalars(n, top, dev, 7)
    ctx.run(sc)
    auto = ctx.last_timings()["window_bits"]
    rows = []
    for c in range(max(2, auto - 3), min(23, auto + 3) + 1):
        ctx.set_option("window_bits", c)
        ctx.run(sc)
        best, tm_best = float("inf"), None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.run(sc)
            dt = (time.perf_counter() - t0) * 1e3
            if dt < best:
                best, tm_best = dt, ctx.last_timings()
        tm = tm_best
        windows, buckets = tm["windows"], tm["windows"] * (1 << (c - 1))
        entries = tm["entries"]
        ns_entry = tm["accumulate"] * 1e6 / entries
        ns_bucket = (tm["segreduce"] + tm["bucket_reduce"]) * 1e6 / buckets
        rows.append((c, best, ns_entry, ns_bucket, (tm["digits"] + tm["sort"]) * 1e6 / entries))
    ctx.close()
    fastest = min(rows, key=lambda r: r[1])
    print("%s 2^%d: model chose c = %d, measured fastest c = %d (%.3f ms vs %.3f ms at the model's choice)"
          % (curve, npow, auto, fastest[0], fastest[1], [r for r in rows if r[0] == auto][0][1]))
    for c, ms, ne, nb, ng in rows:
        print("    c=%2d  %9.3f ms   accumulate %.3f ns/entry   merge+reduce %.3f ns/bucket   grouping %.3f ns/entry   -> BUCKET = %.1f%s"
              % (c, ms, ne, nb, ng, 10.0 * nb / ne, "   <- auto" if c == auto else ""))
        if npow >= 24 and abs(c - auto) <= 1:
            bucket_units.append(10.0 * nb / ne)
    lo = [r for r in rows if r[0] == 21]
    hi = [r for r in rows if r[0] == 22]
    if lo and hi and npow >= 24:
        group_units.append(max(0.0, (hi[0][4] - lo[0][4]) / lo[0][2] / 10.0 * 10.0 / ((257 + 21) // 22)))
if bucket_units:
    print("\n// measured on this box (tools/calibrate_window_model.py %s): paste into choose_window_bits" % curve)
    print("//   bucket cost  %.1f field-multiplication units per bucket (in the code: 50.0)" % (sum(bucket_units) / len(bucket_units)))
if group_units:
    print("//   second grouping pass from 22 bits on: %.3f additions' worth per entry and window (in the code: 0.06)" % (sum(group_units) / len(group_units)))
