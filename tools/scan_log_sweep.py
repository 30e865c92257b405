"""Wall time against the element count per window at which the bucket reduction switches from chunked running sums to the
parallel scan (option "reduce_scan_log"), with the first-level chunk following from it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entries_amd as ea
import bench

curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_377_g1"
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
for npow in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "18,20,22,24,26".split(","))]:
    n = 1 << npow
    bases = tile.repeat(n >> 15, 1).contiguous()
    sc = bench.uniform_scalars(n, bench.R381_TOP if "381" in curve else bench.R377_TOP, dev, 7)
    ctx = ea.MultiScalarMultContext(curve)
    ctx.set_bases(bases)
    row = []
    for sl in (10, 12, 13, 14, 15, 16, 17):
        ctx.set_option("reduce_scan_log", sl)
        for _ in range(2):
            ctx.run(sc)
        ts = []
        for _ in range(7 if npow < 24 else 3):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); ctx.run(sc); ts.append(time.perf_counter() - t0)
        ts.sort()
        tm = ctx.last_timings()
        row.append("%d: %.3f (r %.3f)" % (sl, ts[len(ts) // 2] * 1e3, tm["bucket_reduce"]))
    print("2^%-2d c=%d " % (npow, tm["window_bits"]) + " | ".join(row), flush=True)
    ctx.close()
