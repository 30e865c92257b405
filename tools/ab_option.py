"""Same-box A/B of ONE integer option of a context (mi355_msm_set_option): the values are interleaved round-robin over `reps`
rounds so that clock drift hits all alike; every value's result bytes must equal the first one's.
Usage: [AB_PRE=precompute=2] ab_option.py <option> [curve] [npow] [values] [reps] [window_bits]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entries_amd as ea
import bench

opt = sys.argv.pop(1)
curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_377_g1"
npow = int(sys.argv[2]) if len(sys.argv) > 2 else 26
masks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
wbits = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda", 0)
n = int(os.environ.get("AB_N", 1 << npow))          # AB_N: a size that is not a power of two
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
bases = tile[:n].contiguous() if n <= (1 << 15) else tile.repeat((n >> 15) + 1, 1)[:n].contiguous()
sc = bench.uniform_scalars(n, bench.R381_TOP if "381" in curve else bench.R377_TOP, dev, 7)
ctx = ea.MultiScalarMultContext(curve)
for kv in os.environ.get("AB_PRE", "").split(","):      # options that must be set before the bases, e.g. AB_PRE=precompute=2
    if kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ctx.set_bases(bases)
if wbits:
    ctx.set_option("window_bits", wbits)
del tile
ref = None
acc = {m: [] for m in masks}
stages = {m: None for m in masks}
for m in masks:          # warm every variant once (code upload, buffers)
    ctx.set_option(opt, m)
    out = ctx.run(sc)[0]
    if ref is None:
        ref = out
    assert out == ref, "%s=%d: result differs from %s=%d" % (opt, m, opt, masks[0])
for r in range(reps):
    for m in masks:
        ctx.set_option(opt, m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = ctx.run(sc)[0]
        acc[m].append((time.perf_counter() - t0) * 1e3)
        assert out == ref
        stages[m] = ctx.last_timings()
print("%s 2^%d (n = %d)  c=%s windows=%s  (%d interleaved rounds; wall ms: median [min..max])" % (curve, npow, n, stages[masks[0]]["window_bits"], stages[masks[0]]["windows"], reps))
for m in masks:
    ts = sorted(acc[m])
    tm = stages[m]
    print("%s=%-2d  %8.2f [%7.2f .. %7.2f]   c=%d  accumulate %7.2f  merge %5.2f  bucket_reduce %6.2f  (digits %.2f sort %.2f)" % (
        opt, m, ts[len(ts) // 2], ts[0], ts[-1], tm["window_bits"], tm["accumulate"], tm["segreduce"], tm["bucket_reduce"], tm["digits"], tm["sort"]))
ctx.close()
