"""Wall time of small and medium MSMs against the entries per accumulate lane (K) and the fragment-merge fan-in."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entries_amd as ea
import bench

curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_377_g1"
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)


def wall(ctx, sc):
    for _ in range(3):
        ctx.run(sc)
    ts = []
    for _ in range(9):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ctx.run(sc); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[4] * 1e3


for npow in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "12,14,16,18,20,22,24".split(","))]:
    n = 1 << npow
    bases = tile[:n].contiguous() if n <= (1 << 15) else tile.repeat(n >> 15, 1).contiguous()
    sc = bench.uniform_scalars(n, bench.R381_TOP if "381" in curve else bench.R377_TOP, dev, 7)
    ctx = ea.MultiScalarMultContext(curve)
    ctx.set_bases(bases)
    ctx.run(sc)
    k0 = ctx.last_timings()["lane_entries"]
    row = []
    for k in sorted({int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "4,8,16,32".split(","))} | {max(4, k0 // 2), k0, 2 * k0}):
        ctx.set_option("lane_entries", k)
        row.append("K=%d%s: %.3f" % (k, "*" if k == k0 else "", wall(ctx, sc)))
    ctx.set_option("lane_entries", 0)
    for s in (4, 16):
        ctx.set_option("seg_entries", s)
        row.append("fan-in %d: %.3f" % (s, wall(ctx, sc)))
    print("2^%-2d  " % npow + " | ".join(row), flush=True)
    ctx.close()
