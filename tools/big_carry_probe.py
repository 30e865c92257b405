import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import entries_amd as ea, bench
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1)).to(dev)
for npow in (27, 28):
    n = 1 << npow
    bases = tile.repeat(n >> 15, 1).contiguous()
    sc = bench.uniform_scalars(n, bench.R377_TOP, dev, 5)
    ctx = ea.MultiScalarMultContext("bls12_377_g1")
    ctx.set_bases(bases)
    ctx.run(sc)
    t0 = time.perf_counter(); r1 = ctx.run(sc)[0]; t1 = time.perf_counter() - t0
    tm = ctx.last_timings()
    print("   carried stages:", {k: round(v, 1) for k, v in tm.items() if isinstance(v, float)}, flush=True)
    ctx.set_option("carry", 0)
    ctx.run(sc)
    t0 = time.perf_counter(); r0 = ctx.run(sc)[0]; t2 = time.perf_counter() - t0
    tm0 = ctx.last_timings()
    print("   carry-0 stages:", {k: round(v, 1) for k, v in tm0.items() if isinstance(v, float)}, tm0["launches"], flush=True)
    ctx.close()
    # the same sum as 2^(npow-26) shards folded on the host
    parts = []
    c2 = ea.MultiScalarMultContext("bls12_377_g1")
    for s in range(n >> 26):
        c2.set_bases(bases[s << 26:(s + 1) << 26])
        parts.append(c2.run(sc[s << 26:(s + 1) << 26])[0])
    c2.close()
    ref = ea.fold_partials(parts, "bls12_377_g1")
    print("2^%d one context: carried %.1f ms (c=%d, %d launches) same=%s | carry 0: %.1f ms (c=%d) same=%s" % (
        npow, t1 * 1e3, tm["window_bits"], tm["launches"], r1 == ref, t2 * 1e3, tm0["window_bits"], r0 == ref), flush=True)
    del bases, sc
