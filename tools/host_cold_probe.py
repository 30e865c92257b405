"""On the GPU box: one batch of 2^26 host scalars from memory the runtime has seen before (warm) against freshly allocated
pageable memory (cold: what the first call of a harness hands over)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
warm = sc_dev.cpu().numpy()
ctx.run(warm)
for label, make in (("warm pageable", lambda: warm), ("cold pageable (fresh copy)", lambda: warm.copy()), ("warm pageable", lambda: warm),
                    ("cold pageable (fresh copy)", lambda: warm.copy()), ("pinned", lambda: torch.from_numpy(warm).pin_memory())):
    buf = make()
    t0 = time.perf_counter(); r = ctx.run(buf); dt = (time.perf_counter() - t0) * 1e3
    print("%-28s %.1f ms  same=%s" % (label, dt, r == ref), flush=True)
    del buf
