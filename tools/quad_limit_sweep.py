"""Wall time against the four-lanes-per-addition threshold (option "quad_limit") across sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entries_amd as ea
import bench

dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve="bls12_377_g1")).to(dev)
for npow in (14, 16, 18, 20, 22, 24, 26):
    n = 1 << npow
    bases = tile[:n].contiguous() if n <= (1 << 15) else tile.repeat(n >> 15, 1).contiguous()
    sc = bench.uniform_scalars(n, bench.R377_TOP, dev, 7)
    ctx = ea.MultiScalarMultContext("bls12_377_g1")
    ctx.set_bases(bases)
    row = []
    for lim in (0, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22):
        ctx.set_option("quad_limit", lim)
        for _ in range(2):
            ctx.run(sc)
        ts = []
        for _ in range(7 if npow < 24 else 3):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); ctx.run(sc); ts.append(time.perf_counter() - t0)
        ts.sort()
        tm = ctx.last_timings()
        row.append("2^%d: %.3f (m %.3f r %.3f)" % (lim.bit_length() - 1 if lim else 0, ts[len(ts) // 2] * 1e3, tm["segreduce"], tm["bucket_reduce"]))
    print("2^%-2d  " % npow + " | ".join(row), flush=True)
    ctx.close()
