"""Latency of one MSM (bases and scalars resident) across sizes, with the per-stage device times: where small inputs go."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entries_amd as ea
import bench

curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_377_g1"
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
for npow in (10, 12, 14, 15, 16, 18, 20, 22, 24):
    n = 1 << npow
    bases = tile[:n].contiguous() if n <= (1 << 15) else tile.repeat(n >> 15, 1).contiguous()
    ctx = ea.MultiScalarMultContext(curve)
    ctx.set_bases(bases)
    sc = bench.uniform_scalars(n, bench.R381_TOP if "381" in curve else bench.R377_TOP, dev, 7)
    for _ in range(3):
        ctx.run(sc)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ctx.run(sc); ts.append(time.perf_counter() - t0)
    tm = ctx.last_timings()
    ts.sort()
    print("2^%-2d  wall median %8.3f ms  min %8.3f | device total %7.3f = digits %.3f sort %.3f accumulate %.3f merge %.3f reduce %.3f | c=%d W=%d K=%d lanes=%d" % (
        npow, ts[len(ts) // 2] * 1e3, ts[0] * 1e3, tm["total"], tm["digits"], tm["sort"], tm["accumulate"], tm["segreduce"], tm["bucket_reduce"],
        tm["window_bits"], tm["windows"], tm["lane_entries"], tm["lanes"]), flush=True)
    ctx.close()
