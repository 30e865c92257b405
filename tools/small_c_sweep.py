"""Wall time of small MSMs against the window size c: where the automatic choice leaves a nearly empty top window (253 mod c in
{0, 1}: one bucket holds the 14 % of the scalars with bit 252 set) the fragment merge walks one very long run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entries_amd as ea
import bench

curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_377_g1"
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
for npow in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "10,12,14,15,16,17,18,20".split(","))]:
    n = 1 << npow
    bases = tile[:n].contiguous() if n <= (1 << 15) else tile.repeat(n >> 15, 1).contiguous()
    sc = bench.uniform_scalars(n, bench.R381_TOP if "381" in curve else bench.R377_TOP, dev, 7)
    ctx = ea.MultiScalarMultContext(curve)
    ctx.set_bases(bases)
    ctx.run(sc)
    c0 = ctx.last_timings()["window_bits"]
    row = []
    for c in range(max(2, c0 - 5), c0 + 3):
        ctx.set_option("window_bits", c)
        for _ in range(3):
            ctx.run(sc)
        ts = []
        for _ in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); ctx.run(sc); ts.append(time.perf_counter() - t0)
        ts.sort()
        tm = ctx.last_timings()
        row.append("c=%d%s top=%d: %.3f (merge %.3f red %.3f)" % (c, "*" if c == c0 else "", 253 % c, ts[4] * 1e3, tm["segreduce"], tm["bucket_reduce"]))
    print("2^%-2d  " % npow + " | ".join(row), flush=True)
    ctx.close()
