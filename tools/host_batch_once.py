"""One 2^26 host-scalar batch (for rocprofv3 --kernel-trace --stats: the kernels of a carried batch, k_bucket_merge among them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entries_amd as ea
import bench

n = 1 << 26
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1)).to(dev)
ctx = ea.MultiScalarMultContext("bls12_377_g1")
ctx.set_bases(tile.repeat(n >> 15, 1).contiguous())
sc = bench.uniform_scalars(n, bench.R377_TOP, dev, 7).cpu().numpy()
for _ in range(3):
    ctx.run(sc)
print(ctx.last_timings())
