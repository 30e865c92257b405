"""Extended randomized parity soak (run by hand on the GPU box): N cases over curves, sizes, scalar shapes and knobs."""
import ctypes, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import entries_amd as ea
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pymodel as pm
import te_model as te

lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
lib.oracle_msm.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
CURVES = [("bls12_377_g1", 0, 0x12ab655e9a2ca556), ("bls12_381_g1", 1, 0x73eda753299d7d48), ("bls12_377_g2", 2, 0x12ab655e9a2ca556), ("bls12_381_g2", 3, 0x73eda753299d7d48)]
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
G2_MIX = [2, 2, 2, 3, 3, 3, 0, 1]     # argv[3] == "g2": mostly G2 cases (the paired kernels)
bad = 0
for case in range(cases):
    name, cid, top = CURVES[rng.choice(G2_MIX if len(sys.argv) > 3 and sys.argv[3] == "g2" else [0, 0, 0, 1, 1, 1, 2, 3])]
    n = rng.choice([1, 2, 5, 31, 32, 33, 100, 257, 1023, 1024, 3000, 8191, 8193, 9999, 40000, 70001, 200000])
    if cid >= 2:
        n = min(n, 3000)
    stride = ea.affine_stride(name)
    bases = ea.generate_points(n, distinct=rng.choice([1, 3, 50, n]), seed=case, curve=name)
    for _ in range(rng.randrange(3)):
        bases[rng.randrange(n), stride - 8] = 1
    special = ""
    if cid == 0 and rng.random() < 0.35:
        # BLS12-377 G1 off the prime-order subgroup: points without a twisted-Edwards image (the base set must stay on XYZZ),
        # or mappable points of even order (P + E) whose sums can hit a vanishing denominator (the run must fall back)
        G1 = pm.BLS12_377_G1
        prng = random.Random(case)
        E = prng.choice(te.exceptional_points())
        P = pm.random_points(G1, 1, prng)[0]
        if prng.random() < 0.5:
            pts, special = [E], "exceptional"
        else:
            pts, special = [G1.add(P, E), P, G1.neg(P)], "even-order"
        for Q in pts:
            if te.sw_to_te(Q) is None and special == "even-order":
                continue
            bases[prng.randrange(n)] = np.frombuffer(G1.encode_affine(Q), dtype=np.uint8)
    g = np.random.default_rng(case)
    limbs = g.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    kind = rng.randrange(6)
    if kind < 3:
        limbs[:, 3] %= np.uint64(top)
    elif kind == 3:
        limbs[:, 1:] = 0
    elif kind == 4:
        limbs[:] = limbs[0]
        limbs[:, 3] %= np.uint64(top)
    # kind 5: full 256-bit scalars (exact integer semantics; the oracle follows arkworks which windows all 256 bits for these c)
    sc = limbs.view(np.uint8).reshape(n, 32)
    shards = rng.choice([0, 0, 0, 2, 5])     # 0: ordinary context; else that many logical shards behind the C ABI
    ctx = ea.MultiScalarMultContext(name, devices=[0] * shards) if shards else ea.MultiScalarMultContext(name)
    opts = {"shards": shards} if shards else {}
    if rng.random() < 0.35:
        opts["precompute"] = 1
        if rng.random() < 0.6:
            opts["table_levels"] = rng.choice([2, 3, 4, 6, 9])      # round 4: k levels, ceil(W / k) bucket sets
    elif rng.random() < 0.1:
        opts["precompute"] = 2                                      # round 5: auto (no tables at these sizes: the option must be inert)
    if rng.random() < 0.5:
        opts["window_bits"] = rng.randrange(2, 25) if rng.random() < 0.3 else rng.randrange(2, 18)
    if cid == 0 and rng.random() < 0.25:
        opts["twisted_edwards"] = 0
    for k, v in opts.items():
        if k != "shards":
            ctx.set_option(k, v)
    ctx.set_bases(torch.from_numpy(bases).cuda() if rng.random() < 0.5 else bases)
    if rng.random() < 0.5:
        ctx.set_option("lane_entries", rng.choice([1, 3, 8, 64, 1000])); opts["lane"] = 1
    if rng.random() < 0.3:
        ctx.set_option("seg_entries", rng.choice([4, 6, 33]))
    if rng.random() < 0.3:
        ctx.set_option("max_chunk", rng.choice([211, 4096]))
    if rng.random() < 0.15:
        ctx.set_option("mem_limit", rng.choice([1 << 20, 16 << 20])); opts["mem_limit"] = 1
    if rng.random() < 0.3:
        ctx.set_option("carry", 0); opts["carry"] = 0          # chunks reduce their own buckets (the default carries one bucket array)
    if special == "" and rng.random() < 0.3:
        ctx.set_option("assume_subgroup", 1); opts["fold"] = 1  # the generator's points are multiples of G: scalars above r/2 fold
    if rng.random() < 0.5:
        ctx.set_option("anchor", 2); opts["anchor"] = 2         # round 6: the anchored window at any size and window size (msm_engine.hip)
    if cid >= 2 and rng.random() < 0.7:
        # round 5: which G2 throughput kernels run two lanes per point (csrc/fp2pair.hpp); quad_limit = 0 sends the merge and scan
        # launches of these small inputs through them at all
        opts["g2_paired"] = rng.choice([0, 1, 2, 4, 8, 16, 31, rng.randrange(32)])
        ctx.set_option("g2_paired", opts["g2_paired"])
        if rng.random() < 0.6:
            ctx.set_option("quad_limit", 0); opts["quad"] = 0
    if shards and rng.random() < 0.4:
        ctx.set_option("force_peer_staging", 1); opts["peer"] = 1   # round 5: the cross-device staging branches of a sharded context
    got = ctx.run(torch.from_numpy(sc).cuda() if rng.random() < 0.5 else sc)[0]
    ctx.close()
    if rng.random() < 0.15 and not shards:
        # the stateless call on the same operands (the pipeline, the bounded pool)
        if ea.msm(bases, sc, name) != got:
            bad += 1
            print("MISMATCH stateless vs context", case, name, n, kind, opts, special, flush=True)
    out = ctypes.create_string_buffer(ea.projective_bytes(name))
    lib.oracle_msm(cid, bases.ctypes.data, stride, sc.ctypes.data, n, out, 0)
    if kind == 5:
        continue   # arkworks ignores scalar bits >= MODULUS_BIT_SIZE for some window sizes; covered by the big-int model test
    if got != out.raw:
        bad += 1
        print("MISMATCH", case, name, n, kind, opts, special, flush=True)
print("fuzz done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
