"""On the GPU box: accumulate time against entries per lane K around the default 512 -- the number of ROUNDS the resident blocks need
(working blocks / (256 CUs x 3 blocks)) is what matters near the tail: a last round that is 9 % full idles the chip for a block's duration.
usage: tools/lane_round_sweep.py [curve=bls12_377_g1] [npow=26] [Ks=...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, entries_amd as ea, bench
curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_377_g1"
npow = int(sys.argv[2]) if len(sys.argv) > 2 else 26
Ks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "416,448,464,480,496,512,520,528,544,560,592,640,688").split(",")]
n = 1 << npow
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
ctx = ea.MultiScalarMultContext(curve)
ctx.set_bases(tile.repeat(n >> 15, 1).contiguous())
sc = bench.uniform_scalars(n, bench.R381_TOP if "381" in curve else bench.R377_TOP, dev, 7)
ctx.run(sc)
adds = ctx.query("sorted_entries")
wpb = 2 if curve.endswith("g2") else 1       # G2 pair form: two hardware lanes per walking lane
res = {k: [] for k in Ks}
for r in range(3):
    for k in Ks:
        ctx.set_option("lane_entries", k)
        ctx.run(sc)
        torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.run(sc); wall = (time.perf_counter() - t0) * 1e3
        t = ctx.last_timings()
        res[k].append((t["accumulate"], t["segreduce"], wall))
print("%s 2^%d c=%d, %d mixed additions per launch" % (curve, npow, t["window_bits"], adds))
for k in Ks:
    blocks = adds / k / (256 / wpb)
    a = sorted(x[0] for x in res[k])[1]; m = sorted(x[1] for x in res[k])[1]; w = sorted(x[2] for x in res[k])[1]
    print("K=%4d  working blocks %7.0f = %5.2f rounds of 768   accumulate %7.2f  merge %5.2f  wall %7.2f" % (k, blocks, blocks / 768, a, m, w))
