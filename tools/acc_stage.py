"""One context, BLS12-377 G1 2^26, five runs: the accumulate stage time of each (A/B helper for engine variants: tools/ab_variants.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, entries_amd as ea, bench
npow = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << npow
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve="bls12_377_g1")).to(dev)
ctx = ea.MultiScalarMultContext("bls12_377_g1")
if len(sys.argv) > 2:
    ctx.set_option("window_bits", int(sys.argv[2])); ctx.set_option("anchor", 0 if int(sys.argv[2]) == 20 else 1)
ctx.set_bases(tile.repeat(n >> 15, 1).contiguous())
sc = bench.uniform_scalars(n, bench.R377_TOP, dev, 7)
acc = []
for _ in range(5):
    ctx.run(sc); t = ctx.last_timings(); acc.append(t["accumulate"])
print("c=%d accumulate ms: %s  (te fallbacks %d)" % (t["window_bits"], " ".join("%.2f" % a for a in acc[1:]), ctx.query("twisted_edwards_fallbacks")))
