"""PCIe-inclusive rate: scalars start in host memory (what the Rust harness hands over), bases resident (init untimed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import entries_amd as ea
import bench

npow = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << npow
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1)).to(dev)
ctx = ea.MultiScalarMultContext("bls12_377_g1")
ctx.set_bases(tile.repeat(n >> 15, 1).contiguous())
for batches in (1, 4):
    sc_dev = bench.uniform_scalars(n * batches, bench.R377_TOP, dev, 7)
    sc_host = sc_dev.cpu().numpy()          # pageable host memory
    sc_pinned = sc_dev.cpu().pin_memory().numpy()
    for name, buf in (("device", sc_dev), ("host pageable", sc_host), ("host pinned", sc_pinned)):
        ctx.run(buf)
        t0 = time.perf_counter(); r = ctx.run(buf); dt = time.perf_counter() - t0
        print("2^%d x %d batches, scalars %-14s: %.1f ms total, %.1f ms per MSM" % (npow, batches, name, dt * 1e3, dt * 1e3 / batches), flush=True)
    del sc_dev
