#!/usr/bin/env python3
"""On the GPU box: ms per 2^26-pair BLS12-377 G1 MSM (scalars resident) for table_levels k x window bits c; prints one row each.
usage: tools/table_levels_sweep.py [npow=26]"""
import sys, time
import torch
sys.path.insert(0, ".")
import entries_amd as ea
from bench import uniform_scalars, R377_TOP

npow = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n, distinct = 1 << npow, 1 << 15
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(distinct, distinct=distinct, seed=0x5A5052495A45)).to(dev)
scalars = uniform_scalars(n, R377_TOP, dev, seed=1234)
ref = None
for levels, cs in ((-1, (20,)), (0, (24, 23)), (6, (21, 22, 23, 24)), (4, (22, 23)), (3, (21, 22, 23)), (2, (21, 22))):
    for c in cs:
        try:
            ctx = ea.MultiScalarMultContext("bls12_377_g1")
            if levels >= 0:
                ctx.set_option("precompute", 1)
                ctx.set_option("table_levels", levels)
            ctx.set_option("window_bits", c)
            t0 = time.perf_counter()
            ctx.set_bases(tile.repeat(n // distinct, 1).contiguous())
            torch.cuda.synchronize()
            init = time.perf_counter() - t0
            r = ctx.run(scalars)[0]
            ref = ref or r
            t0 = time.perf_counter()
            for _ in range(4):
                r = ctx.run(scalars)[0]
            ms = (time.perf_counter() - t0) / 4 * 1e3
            tm = ctx.last_timings()
            print("levels %2d c %2d: %7.2f ms  (windows %2d, table levels %2d, bucket sets %2d, tables %5.1f GB, init %4.1f s; digits %.2f sort %.2f accumulate %.2f reduce %.2f) same=%s"
                  % (levels, c, ms, tm["windows"], ctx.query("table_levels"), -(-tm["windows"] // max(1, ctx.query("table_levels"))), ctx.query("base_bytes") / 1e9, init,
                     tm["digits"], tm["sort"], tm["accumulate"], tm["bucket_reduce"], r == ref), flush=True)
            ctx.close()
        except Exception as e:  # noqa: BLE001
            print("levels %2d c %2d: %r" % (levels, c, e), flush=True)
