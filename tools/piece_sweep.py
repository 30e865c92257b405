"""On the GPU box: the first piece of a host-scalar batch (n/div, then x 3) against div, carried buckets (profiles/r03_ab_carry.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entries_amd as ea
import bench

n = 1 << 26
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1)).to(dev)
ctx = ea.MultiScalarMultContext("bls12_377_g1")
ctx.set_bases(tile.repeat(n >> 15, 1).contiguous())
sc_dev = bench.uniform_scalars(n, bench.R377_TOP, dev, 7)
ref = ctx.run(sc_dev)
host = sc_dev.cpu().numpy()
for rnd in (1, 2):
    for div in (13, 9, 20, 27, 40, 6):
        ctx.set_option("first_piece_div", div)
        ctx.run(host)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); r = ctx.run(host); ts.append((time.perf_counter() - t0) * 1e3)
        print("round %d  first piece 1/%-2d: %.1f ms (min of 4; %d chunks)  same=%s" % (rnd, div, min(ts), ctx.last_timings()["launches"], r == ref), flush=True)
