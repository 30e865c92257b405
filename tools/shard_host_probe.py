#!/usr/bin/env python3
"""Host-scalar delivery to a sharded context (verdict r2 item 3c): BASELINE configs[3] -- 2^28 pairs as 8 shards of 2^25 -- with
the scalars in HOST memory, every shard pulling its slice through its own pinned staging + copy stream, against the same run
with the scalars already on the device.  On the one GPU of the test box the 8 logical shards share ONE PCIe link and ONE set
of CUs (so compute serialises: 8 x T(2^25)); what the probe isolates is the delivery: (host - device) is the part of the
8.6 GB upload that did NOT hide behind the shards' compute.  On an 8-GPU node every shard has its own link and the budget per
link is 2^25 x 32 B = 1.07 GB = ~19 ms at 57 GB/s against ~57 ms of compute.   usage: tools/shard_host_probe.py [total_npow] [shards]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench
import entries_amd as ea

tp = int(sys.argv[1]) if len(sys.argv) > 1 else 28
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 1 << tp
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1)).to(dev)
bases = tile.repeat(n >> 15, 1).contiguous()
ctx = ea.MultiScalarMultContext("bls12_377_g1", devices=[0] * G)
ctx.set_bases(bases)
del bases
torch.cuda.empty_cache()
d_sc = bench.uniform_scalars(n, bench.R377_TOP, dev, 11)
h_sc = d_sc.cpu().numpy()                     # pageable host memory, 32 B x n
pin = torch.empty((n, 32), dtype=torch.uint8).pin_memory()
pin.copy_(torch.from_numpy(h_sc))
print("2^%d pairs, %d logical shards of 2^%.2f on one MI355X; scalars %.2f GB; PCIe floor at 57 GB/s = %.1f ms" %
      (tp, G, np.log2(n / G), n * 32 / 1e9, n * 32 / 57e9 * 1e3), flush=True)


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps * 1e3, r


t_dev, r_dev = timed(lambda: ctx.run(d_sc)[0])
per = ctx.shard_timings()
t_host, r_host = timed(lambda: ctx.run(h_sc)[0])
t_pin, r_pin = timed(lambda: ctx.run(pin.numpy())[0])
print("scalars on the device     %8.1f ms" % t_dev)
print("scalars in host memory    %8.1f ms  (pageable)   same=%s   delivery not hidden: %+.1f ms" % (t_host, r_host == r_dev, t_host - t_dev))
print("scalars in pinned memory  %8.1f ms               same=%s   delivery not hidden: %+.1f ms" % (t_pin, r_pin == r_dev, t_pin - t_dev))
print("per-shard device totals (ms):", " ".join("%.1f" % s.get("total", 0.0) for s in per))
ctx.close()
