#!/usr/bin/env python3
"""On the GPU box: do precomputed tables pay BELOW 2^24 pairs?  (reference shape: CMB PrecomputePoints.cu:10-39,
MSM.cu:380-383 -- tables built in the untimed init.)  For each curve and size: the table-free context against contexts with a table
level per window, 6 and 3 levels (window size = the engine's own choice for that shape); wall ms per MSM (median of 7, scalars
resident), the stage times, table bytes and init seconds.  One row per (curve, size, shape); `best` marks the winner and what
"precompute" = auto picks is printed last.
usage: tools/size_sweep_tables.py [curves=bls12_377_g1,bls12_381_g1,bls12_377_g2] [npows=18,19,20,21,22,23,24]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entries_amd as ea
import bench

curves = (sys.argv[1] if len(sys.argv) > 1 else "bls12_377_g1,bls12_381_g1,bls12_377_g2").split(",")
npows = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "18,19,20,21,22,23,24").split(",")]
dev = torch.device("cuda", 0)


def measure(ctx, sc, reps=7):
    for _ in range(2):
        r = ctx.run(sc)[0]
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.run(sc)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3, r, ctx.last_timings()


for curve in curves:
    tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
    top = bench.R381_TOP if "381" in curve else bench.R377_TOP
    for npow in npows:
        if "g2" in curve and npow > 22 and len(sys.argv) <= 2:
            continue
        n = 1 << npow
        bases = tile[:n].contiguous() if n <= (1 << 15) else tile.repeat(n >> 15, 1).contiguous()
        sc = bench.uniform_scalars(n, top, dev, 7)
        rows, ref = [], None
        for name, pre, lv in (("none", 0, 0), ("all", 1, 0), ("6", 1, 6), ("3", 1, 3), ("auto", 2, 0)):
            ctx = ea.MultiScalarMultContext(curve)
            try:
                if pre:
                    ctx.set_option("precompute", pre)
                    if lv:
                        ctx.set_option("table_levels", lv)
                t0 = time.perf_counter()
                ctx.set_bases(bases)
                torch.cuda.synchronize()
                init = time.perf_counter() - t0
                ms, r, tm = measure(ctx, sc)
                ref = ref or r
                rows.append((name, ms, tm, ctx.query("table_levels"), ctx.query("base_bytes") / 1e9, init, r == ref))
            except Exception as e:  # noqa: BLE001
                rows.append((name, float("inf"), None, 0, 0.0, 0.0, repr(e)))
            finally:
                ctx.close()
        best = min(rows[:4], key=lambda x: x[1])[0]
        base_ms = rows[0][1]
        for name, ms, tm, levels, gb, init, same in rows:
            if tm is None:
                print("%-13s 2^%-2d %-5s %s" % (curve, npow, name, same), flush=True)
                continue
            print("%-13s 2^%-2d %-5s %8.3f ms (%+6.1f %%)  c=%2d W=%2d levels=%2d  tables %6.2f GB init %5.2f s | digits %.2f sort %.2f accumulate %.2f merge %.2f reduce %.2f  same=%s%s"
                  % (curve, npow, name, ms, 100.0 * (ms / base_ms - 1.0), tm["window_bits"], tm["windows"], levels, gb, init, tm["digits"], tm["sort"], tm["accumulate"],
                     tm["segreduce"], tm["bucket_reduce"], same, "   <- best" if name == best else ""), flush=True)
