"""On the GPU box: the anchored window at the sizes where it is the DEFAULT (>= 2^20 pairs), against the plain-digit path of the same
context (which the oracle pins at the sizes it finishes): random curve, ragged n in [2^20, 2^23], tables (auto) or not, chunked or not,
host or device scalars, one or two batches, scalars with zeros / ones / non-canonical values sprinkled in.  Bytes must be equal.
usage: tools/anchor_soak.py [cases=40] [seed=1]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import entries_amd as ea

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
CURVES = ["bls12_377_g1", "bls12_381_g1", "bls12_377_g2", "bls12_381_g2"]
dev = torch.device("cuda", 0)
tiles = {c: torch.from_numpy(ea.generate_points(1 << 14, distinct=1 << 14, seed=3, curve=c)).to(dev) for c in CURVES}
bad = anchored = 0
t0 = time.time()
for case in range(cases):
    curve = rng.choice(CURVES if rng.random() < 0.7 else CURVES[:2])
    g2 = curve.endswith("g2")
    n = rng.randrange(1 << 20, (1 << 22) if g2 else (1 << 23) + (1 << 21))
    reps = -(-n // (1 << 14))
    bases = tiles[curve].repeat(reps, 1)[:n].contiguous()
    for _ in range(rng.randrange(3)):
        bases[rng.randrange(n), -8] = 1                     # a base at infinity
    batches = rng.choice([1, 1, 2])
    nprng = np.random.default_rng(case)
    sc = nprng.integers(0, 256, size=(batches * n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x0F if rng.random() < 0.5 else (0x1F if "377" in curve else 0x7F)   # canonical-ish / up to 2^253 (2^255)
    for _ in range(200):
        sc[rng.randrange(batches * n)] = 0
    sc[rng.randrange(batches * n), :] = 255                 # 2^256 - 1
    sc[rng.randrange(batches * n), 1:] = 0
    opts = {}
    ctx = ea.MultiScalarMultContext(curve)
    if rng.random() < 0.35:
        ctx.set_option("precompute", 2); opts["precompute"] = 2
    if rng.random() < 0.3:
        wb = rng.choice([17, 18, 21, 23] if "377" in curve else [15, 17]); ctx.set_option("window_bits", wb); opts["window_bits"] = wb
    ctx.set_bases(bases)
    if rng.random() < 0.4:
        mc = rng.randrange(n // 5, n); ctx.set_option("max_chunk", mc); opts["max_chunk"] = mc
    operand = torch.from_numpy(sc).to(dev) if rng.random() < 0.5 else sc
    opts["scalars"] = "device" if isinstance(operand, torch.Tensor) else "host"
    ctx.set_option("anchor", 0)
    ref = ctx.run(operand)
    ctx.set_option("anchor", 1)
    got = ctx.run(operand)
    aw, c = ctx.query("anchored_window"), ctx.last_timings()["window_bits"]
    anchored += aw > 0
    if got != ref:
        bad += 1
        print("MISMATCH", case, curve, n, batches, opts, "c", c, "anchored", aw, flush=True)
    ctx.close()
    del bases, operand
print("anchor soak done: %d cases (%d ran with an anchored window), %d mismatches, %.0f s" % (cases, anchored, bad, time.time() - t0))
sys.exit(1 if bad else 0)
