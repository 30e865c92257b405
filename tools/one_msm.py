import sys, os
sys.path.insert(0, "/root/repo")
import torch, entries_amd as ea, bench
npow = int(sys.argv[1]); n = 1 << npow
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(min(n, 1 << 15), distinct=min(n, 1 << 15), seed=1, curve="bls12_377_g1")).to(dev)
bases = tile if n <= (1 << 15) else tile.repeat(n >> 15, 1).contiguous()
sc = bench.uniform_scalars(n, bench.R377_TOP, dev, 7)
ctx = ea.MultiScalarMultContext("bls12_377_g1"); ctx.set_bases(bases)
import time
for _ in range(5):
    ctx.run(sc); torch.cuda.synchronize(); time.sleep(0.002)
