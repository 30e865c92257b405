"""G2 2^24 with precomputed tables (option precompute = 2 / 1 + table_levels) against the default context: what tables are worth for G2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entries_amd as ea
import bench

curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_377_g2"
npow = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda", 0)
n = 1 << npow
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
bases = tile.repeat(n >> 15, 1).contiguous()
sc = bench.uniform_scalars(n, bench.R381_TOP if "381" in curve else bench.R377_TOP, dev, 7)
ref = None
for name, opts in (("default", {}), ("auto", {"precompute": 2}), ("levels6", {"precompute": 1, "table_levels": 6}), ("levels3", {"precompute": 1, "table_levels": 3})):
    ctx = ea.MultiScalarMultContext(curve)
    for k, v in opts.items():
        ctx.set_option(k, v)
    t0 = time.perf_counter()
    ctx.set_bases(bases)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    out = ctx.run(sc)[0]
    ref = ref or out
    assert out == ref, name
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ctx.run(sc); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    tm = ctx.last_timings()
    print("%-8s %7.2f ms  init %.1f s  levels %d c=%d windows=%d bucket_sets=%d table GB %.1f | digits %.2f sort %.2f accumulate %.2f merge %.2f reduce %.2f" % (
        name, ts[2], t_init, ctx.query("table_levels"), tm["window_bits"], tm["windows"], ctx.query("bucket_windows"), ctx.query("base_bytes") / 1e9,
        tm["digits"], tm["sort"], tm["accumulate"], tm["segreduce"], tm["bucket_reduce"]), flush=True)
    ctx.close()
