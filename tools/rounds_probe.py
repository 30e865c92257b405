"""On the GPU box: is it the number of block ROUNDS (working blocks / resident blocks) or K itself?  Fixed K = 512, the number of pairs varied so that
the accumulate launch needs 7.6 ... 8.7 rounds of 768 blocks; ns per mixed addition.  usage: tools/rounds_probe.py [curve] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, entries_amd as ea, bench
curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_377_g1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
per_pair = 12.143 if "377" in curve else 13.0
rows = []
for rounds in [float(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "6.55,6.70,6.85,6.95,7.05,7.15,7.30,7.45,7.60,7.80,7.95,8.05").split(",")]:
    n = int(rounds * 768 * 256 * K / per_pair)
    ctx = ea.MultiScalarMultContext(curve)
    ctx.set_option("window_bits", 21 if "377" in curve else 20)
    ctx.set_option("lane_entries", K)
    ctx.set_bases(tile.repeat((n >> 15) + 1, 1)[:n].contiguous())
    sc = bench.uniform_scalars(n, bench.R381_TOP if "381" in curve else bench.R377_TOP, dev, 7)
    ctx.run(sc)
    acc = []
    for _ in range(4):
        ctx.run(sc); acc.append(ctx.last_timings()["accumulate"])
    adds = ctx.query("sorted_entries")
    a = sorted(acc)[1]
    print("n=%9d  adds %10d  blocks %6.0f = %5.2f rounds   accumulate %7.2f ms  %.4f ns/add" % (n, adds, adds / K / 256, adds / K / 256 / 768, a, a * 1e6 / adds), flush=True)
    ctx.close()
    del sc
