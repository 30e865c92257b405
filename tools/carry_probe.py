"""On the GPU box: what carried buckets buy where a batch runs as several chunks (profiles/r03_ab_carry.txt).
  * one host-scalar batch of 2^26 (the first piece is computed while the rest crosses PCIe): piece = 1/4, 1/8, 1/16 of the batch,
    carried or not (carry = 0: the piece is reduced on its own and added on the host, the round-2 path)
  * four host-scalar batches (the ZPrize workload)
  * the stateless call (slices of one carried batch vs slices that reduce their own buckets: MI355_MSM_STATELESS_CARRY=0)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import entries_amd as ea
import bench

npow = int(sys.argv[1]) if len(sys.argv) > 1 else 26
curve = sys.argv[2] if len(sys.argv) > 2 else "bls12_377_g1"
n = 1 << npow
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
ctx = ea.MultiScalarMultContext(curve)
bases = tile.repeat(n >> 15, 1).contiguous()
ctx.set_bases(bases)
top = bench.R381_TOP if "381" in curve else bench.R377_TOP
sc_dev = bench.uniform_scalars(4 * n, top, dev, 7)
sc_host = sc_dev.cpu().numpy()
ref = ctx.run(sc_dev)


def timed(buf, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = ctx.run(buf); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r

ms, _ = timed(sc_dev[:n])
print("device-resident scalars, one batch: %.1f ms" % ms, flush=True)
for carry in (1, 0):
    for div in (4, 13, 40, 8):
        ctx.set_option("carry", carry)
        ctx.set_option("first_piece_div", div)
        ms1, r1 = timed(sc_host[:n])
        ms4, r4 = timed(sc_host, reps=2)
        t = ctx.last_timings()
        print("carry %d  first piece 1/%-2d: one batch %.1f ms   four batches %.1f ms   same=%s  (c=%d)" % (carry, div, ms1, ms4, r1[0] == ref[0] and r4 == ref, t["window_bits"]), flush=True)
ctx.set_option("carry", 1)
ctx.set_option("first_piece_div", 0)
ctx.close()
del ctx
bases_host = bases.cpu().numpy()
del bases
for env in ("1", "0"):
    os.environ["MI355_MSM_STATELESS_CARRY"] = env
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); r = ea.msm(bases_host, sc_host[:n], curve); best = min(best, time.perf_counter() - t0)
    st = ea.last_stateless() if hasattr(ea, "last_stateless") else {}
    print("stateless, MI355_MSM_STATELESS_CARRY=%s: %.1f ms  same=%s  %s" % (env, best * 1e3, r == ref[0], st), flush=True)
