"""Same-box A/B of the G2 throughput kernels: one Fp2 point per lane (one wave per SIMD) against a point spread over two lanes
(csrc/fp2pair.hpp, two waves per SIMD), per kernel via the bit mask of option "g2_paired":
  1 accumulate, 2 first level of the bucket reduction, 4 fragment merge, 8 scan steps, 16 bucket merge (carried batches).
Variants are interleaved (round-robin, `reps` rounds) so that clock drift hits all alike; every variant's result bytes must equal
variant 0's.  Usage: ab_g2_paired.py [curve] [npow] [masks] [reps] [window_bits]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import entries_amd as ea
import bench

curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_377_g2"
npow = int(sys.argv[2]) if len(sys.argv) > 2 else 24
masks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2,3,15").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
wbits = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda", 0)
n = 1 << npow
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
bases = tile[:n].contiguous() if n <= (1 << 15) else tile.repeat(n >> 15, 1).contiguous()
sc = bench.uniform_scalars(n, bench.R381_TOP if "381" in curve else bench.R377_TOP, dev, 7)
ctx = ea.MultiScalarMultContext(curve)
ctx.set_bases(bases)
if wbits:
    ctx.set_option("window_bits", wbits)
del tile
ref = None
acc = {m: [] for m in masks}
stages = {m: None for m in masks}
for m in masks:          # warm every variant once (code upload, buffers)
    ctx.set_option("g2_paired", m)
    out = ctx.run(sc)[0]
    if ref is None:
        ref = out
    assert out == ref, "g2_paired=%d: result differs from g2_paired=%d" % (m, masks[0])
for r in range(reps):
    for m in masks:
        ctx.set_option("g2_paired", m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = ctx.run(sc)[0]
        acc[m].append((time.perf_counter() - t0) * 1e3)
        assert out == ref
        stages[m] = ctx.last_timings()
print("%s 2^%d  c=%s windows=%s  (%d interleaved rounds; wall ms: median [min..max])" % (curve, npow, stages[masks[0]]["window_bits"], stages[masks[0]]["windows"], reps))
for m in masks:
    ts = sorted(acc[m])
    tm = stages[m]
    print("g2_paired=%-2d  %8.2f [%7.2f .. %7.2f]   accumulate %7.2f  merge %5.2f  bucket_reduce %6.2f  (digits %.2f sort %.2f)" % (
        m, ts[len(ts) // 2], ts[0], ts[-1], tm["accumulate"], tm["segreduce"], tm["bucket_reduce"], tm["digits"], tm["sort"]))
ctx.close()
