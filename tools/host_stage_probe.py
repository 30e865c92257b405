import sys, time
sys.path.insert(0, "/root/repo")
import torch, entries_amd as ea, bench
n = 1 << 26
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1)).to(dev)
ctx = ea.MultiScalarMultContext("bls12_377_g1"); ctx.set_bases(tile.repeat(n >> 15, 1).contiguous())
sc = bench.uniform_scalars(n, bench.R377_TOP, dev, 7); sc_h = sc.cpu().numpy(); scp = torch.from_numpy(sc_h).pin_memory()
for name, s in (("device", sc), ("pageable", sc_h), ("pinned", scp)):
    for div in ((0,) if name == "device" else (13, 26)):
        ctx.set_option("first_piece_div", div)
        ctx.run(s)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.run(s); dt = (time.perf_counter() - t0) * 1e3
            if dt < best: best, tm = dt, ctx.last_timings()
        print("%-9s div %2d: %7.2f ms  launches %d  digits %.2f sort %.2f acc %.2f merge %.2f reduce %.2f total(dev) %.2f" % (name, div, best, tm["launches"], tm["digits"], tm["sort"], tm["accumulate"], tm["segreduce"], tm["bucket_reduce"], tm["total"]))
