import sys, os, time
sys.path.insert(0, "/root/repo")
import torch, entries_amd as ea, bench
dev = torch.device("cuda", 0)
for npow in (10, 16):
    n = 1 << npow
    tile = torch.from_numpy(ea.generate_points(min(n, 1 << 15), distinct=min(n, 1 << 15), seed=1, curve="bls12_377_g1")).to(dev)
    bases = tile if n <= (1 << 15) else tile.repeat(n >> 15, 1).contiguous()
    sc = bench.uniform_scalars(n, bench.R377_TOP, dev, 7)
    ctx = ea.MultiScalarMultContext("bls12_377_g1"); ctx.set_bases(bases)
    for _ in range(5): ctx.run(sc)
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.run(sc); ts.append(time.perf_counter() - t0)
    ts.sort(); tm = ctx.last_timings()
    print(npow, "wall %.3f" % (ts[10] * 1e3), {k: round(v, 3) for k, v in tm.items() if isinstance(v, float)})
