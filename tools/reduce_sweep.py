"""GPU: bucket->window reduction time at 2^npow for different first-level chunk sizes (option reduce_log_chunk)."""
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import entries_amd as ea
from bench import uniform_scalars, R377_TOP

npow = int(sys.argv[1]) if len(sys.argv) > 1 else 26
curve = sys.argv[2] if len(sys.argv) > 2 else "bls12_377_g1"
n, distinct = 1 << npow, 1 << 15
tile = torch.from_numpy(ea.generate_points(distinct, distinct=distinct, seed=7, curve=curve)).cuda()
ctx = ea.MultiScalarMultContext(curve)
ctx.set_bases(tile.repeat(n // distinct, 1).contiguous())
sc = uniform_scalars(n, R377_TOP, torch.device("cuda"), 5)
ref = None
for rl in (0, 4, 5, 6, 7):
    ctx.set_option("reduce_log_chunk0", rl)
    ctx.run(sc)
    t0 = time.perf_counter()
    for _ in range(3):
        r = ctx.run(sc)[0]
    dt = (time.perf_counter() - t0) / 3 * 1e3
    tm = ctx.last_timings()
    ref = ref or r
    print("reduce_log_chunk0=%d (later levels: chunks of 4)  step %.2f ms  bucket_reduce %.3f  segreduce %.3f  accumulate %.2f  same=%s" % (rl, dt, tm["bucket_reduce"], tm["segreduce"], tm["accumulate"], r == ref))
