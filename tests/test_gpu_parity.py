"""GPU: parity of the HIP path, called through the C ABI, against the CPU oracle and the golden vectors; then, at
BASELINE.json's full sizes where the oracle is too slow, size-independent properties (linearity in the scalars,
shard-and-fold consistency, the all-ones checksum).  Bit-exact everywhere: results are normalised projective images."""
import ctypes
import os
import random

import numpy as np
import pytest

import pymodel as m
from conftest import oracle_msm, oracle_msm_np

pytestmark = pytest.mark.gpu

CURVES = [(0, m.BLS12_377_G1), (1, m.BLS12_381_G1)]
R_TOP = {0: 0x12ab655e9a2ca556, 1: 0x73eda753299d7d48}


def rand_scalars_np(cid, n, seed):
    """uniform below r (top limb below r's top limb), 4 x u64 LE as uint8[n,32]"""
    rng = np.random.default_rng(seed)
    limbs = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    top = R_TOP[cid]
    limbs[:, 3] %= np.uint64(top)
    return limbs.view(np.uint8).reshape(n, 32)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def test_native_library_is_the_one_running(ea, torch_cuda):
    """No silent fallback: the process has mapped the in-tree HIP library with gfx950 kernels."""
    ea.load_library()
    maps = open("/proc/self/maps").read()
    assert "libmi355msm.so" in maps
    assert "gfx950" in torch_cuda.cuda.get_device_properties(0).gcnArchName


def test_golden_vectors(ea, golden, torch_cuda):
    assert {c["curve"] for c in golden} == {"bls12_377_g1", "bls12_381_g1", "bls12_377_g2", "bls12_381_g2"}
    for case in golden:
        bases, scalars = bytes.fromhex(case["bases"]), bytes.fromhex(case["scalars"])
        ctx = ea.multi_scalar_mult_init(bases, case["curve"])
        got = ea.multi_scalar_mult(ctx, bases, scalars)[0]
        ctx.close()
        assert got.hex() == case["expected"], f'{case["curve"]}/{case["name"]}'


def test_golden_vectors_large(ea, torch_cuda):
    """tests/golden/msm_vectors_large.json: 2^10 and 2^12 pairs on all four curves (cross-checked against every reference build
    at generation, tools/gen_golden.py), through the context path, the stateless call, and -- so that a committed fixture
    crosses a chunk boundary -- with max_chunk forcing three chunks over carried buckets."""
    import json

    from conftest import ROOT
    from test_oracle import _expand_large

    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "msm_vectors_large.json")))["cases"]
    assert len(cases) == 8
    for case in cases:
        bases, scalars = _expand_large(case)
        ctx = ea.multi_scalar_mult_init(bases, case["curve"])
        assert ea.multi_scalar_mult(ctx, bases, scalars)[0].hex() == case["expected"], (case["curve"], case["n"])
        ctx.set_option("max_chunk", case["n"] // 3 + 1)
        assert ctx.run(scalars)[0].hex() == case["expected"], (case["curve"], case["n"], "three chunks")
        ctx.close()
        assert ea.msm(bases, scalars, case["curve"]).hex() == case["expected"], (case["curve"], case["n"], "stateless")


@pytest.mark.parametrize("group,name", [("g1", "bls12_381_g1"), ("g2", "bls12_381_g2")])
def test_rfc9380_vectors_held_by_the_reference(ea, torch_cuda, group, name):
    """The reference-held BLS12-381 G1 / G2 known answers (tests/golden/h2c_kat_bls12_381.json, from the RFC 9380 vectors in
    ARK ec/src/hashing/tests/testdata): P = h_eff (Q0 + Q1) as an MSM through the HIP path; inputs and output are literals of
    the reference, and Q0, Q1 lie OUTSIDE the order-r subgroup."""
    import json

    from conftest import ROOT
    from test_oracle import _h2c_point

    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "h2c_kat_bls12_381.json")))
    c = m.CURVES[name]
    chunks = [int(h, 16) for h in kat[group]["h_eff_chunks"]]
    shift = kat["chunk_bits"]
    all_pts, all_sc, total = [], [], None
    for v in kat[group]["vectors"]:
        q0, q1, P = (_h2c_point(c, v[k]) for k in ("Q0", "Q1", "P"))
        pts, sc = [], []
        for j, h in enumerate(chunks):
            pts += [c.mul(1 << (shift * j), q0), c.mul(1 << (shift * j), q1)]
            sc += [h, h]
        assert ea.msm(c.encode_affine_array(pts), m.encode_scalars(sc), name) == c.encode_projective_normalized(P), v["msg"]
        all_pts += pts
        all_sc += sc
        total = c.add(total, P)
    assert ea.msm(c.encode_affine_array(all_pts), m.encode_scalars(all_sc), name) == c.encode_projective_normalized(total)
    # every scalar split into eight non-negative integer shares (exact over the integers: Q0, Q1 are not in the r-torsion, so
    # shares may not be reduced modulo r): 8x the pairs, the same literal total
    rng = random.Random(9380)
    shares_pts, shares_sc = [], []
    for P, k in zip(all_pts, all_sc):
        cut = sorted(rng.randrange(k + 1) for _ in range(7))
        parts = [b - a for a, b in zip([0] + cut, cut + [k])]     # eight non-negative integers summing to k exactly
        shares_pts += [P] * 8
        shares_sc += parts
    assert ea.msm(c.encode_affine_array(shares_pts), m.encode_scalars(shares_sc), name) == c.encode_projective_normalized(total)


@pytest.mark.parametrize("cid,curve", CURVES)
@pytest.mark.parametrize("npow", [10, 14, 16, 18])
def test_random_vs_oracle(ea, oracle, torch_cuda, cid, curve, npow):
    """msm_correctness (P1A combined-top-solutions/tests/msm.rs:15-40): accelerator == CPU MSM on random inputs,
    2^15-style replicated bases, uniform scalars, several batches on one context."""
    n = 1 << npow
    bases = ea.generate_points(n, distinct=min(n, 1 << 11), seed=npow, curve=curve.name)
    batches = 2 if npow <= 16 else 1
    scalars = rand_scalars_np(cid, n * batches, seed=100 + npow)
    ctx = ea.multi_scalar_mult_init(torch_cuda.from_numpy(bases).cuda(), curve.name)
    got = ea.multi_scalar_mult(ctx, None, torch_cuda.from_numpy(scalars).cuda())
    assert len(got) == batches
    for b in range(batches):
        exp = oracle_msm_np(oracle, cid, bases, np.ascontiguousarray(scalars[b * n:(b + 1) * n]), n)
        assert got[b] == exp, f"{curve.name} 2^{npow} batch {b}"
    # host-pointer entry points give the same bytes
    ctx2 = ea.multi_scalar_mult_init(bases, curve.name)
    assert ea.multi_scalar_mult(ctx2, bases, scalars) == got
    ctx.close()
    ctx2.close()


@pytest.mark.parametrize("cid,curve", CURVES)
def test_ragged_sizes_and_stateless_call(ea, oracle, torch_cuda, cid, curve):
    rng = random.Random(11 + cid)
    for n in (0, 1, 2, 3, 31, 32, 33, 255, 1000, 4097):
        pts = m.random_points(curve, n, rng, max(1, n // 5)) if n else []
        sc = m.random_scalars(curve, n, rng)
        bases, scalars = curve.encode_affine_array(pts), m.encode_scalars(sc)
        assert ea.msm(bases, scalars, curve.name) == oracle_msm(oracle, cid, bases, scalars, n), n
    # VariableBaseMSM::msm chops to the shorter slice; msm_checked reports the shorter length
    pts = m.random_points(curve, 20, rng)
    sc = m.random_scalars(curve, 12, rng)
    bases, scalars = curve.encode_affine_array(pts), m.encode_scalars(sc)
    v = ea.VariableBaseMSM(curve.name)
    assert v.msm(bases, scalars) == oracle_msm(oracle, cid, bases, scalars, 12)
    assert v.msm_checked(bases, scalars) == 12


@pytest.mark.parametrize("cid,curve", CURVES)
def test_skewed_scalar_distributions(ea, oracle, torch_cuda, cid, curve):
    """Hot buckets: equal scalars, tiny scalars, unit and zero scalars, top bits set, all-ones 256-bit values."""
    n = 5000
    bases = ea.generate_points(n, distinct=64, seed=3, curve=curve.name)
    ctx = ea.multi_scalar_mult_init(bases, curve.name)
    rng = np.random.default_rng(5)
    variants = {
        "all_equal": np.tile(rand_scalars_np(cid, 1, 9), (n, 1)),
        "all_one": np.tile(np.array([1] + [0] * 31, dtype=np.uint8), (n, 1)),
        "all_zero": np.zeros((n, 32), dtype=np.uint8),
        "tiny": np.concatenate([rng.integers(0, 4, size=(n, 1), dtype=np.uint8), np.zeros((n, 31), dtype=np.uint8)], axis=1),
        "two_values": np.where(rng.integers(0, 2, size=(n, 1)) == 1, rand_scalars_np(cid, 1, 1), rand_scalars_np(cid, 1, 2)).astype(np.uint8),
        "r_minus_1": np.tile(np.frombuffer((curve.r - 1).to_bytes(32, "little"), dtype=np.uint8), (n, 1)),
    }
    for name, sc in variants.items():
        sc = np.ascontiguousarray(sc)
        got = ea.multi_scalar_mult(ctx, bases, sc)[0]
        assert got == oracle_msm_np(oracle, cid, bases, sc, n), name
    # full 256-bit scalars are exact integers for us; compare with the big-int model (arkworks ignores bits above the modulus size)
    pts = [curve.decode_affine(bases[i].tobytes()) for i in range(40)]
    ks = [(1 << 256) - 1 - i for i in range(40)]
    got = ea.msm(bases[:40].tobytes(), m.encode_scalars(ks), curve.name)
    assert got == curve.encode_projective_normalized(curve.msm_naive(pts, ks))
    ctx.close()


@pytest.mark.parametrize("cid,curve", CURVES)
def test_tuning_knobs_do_not_change_results(ea, oracle, torch_cuda, cid, curve):
    """window size, entries per lane, fragment fan-in and chunking are performance knobs only."""
    n = 1 << 13
    bases = ea.generate_points(n, distinct=300, seed=8, curve=curve.name)
    scalars = rand_scalars_np(cid, n, 77)
    exp = oracle_msm_np(oracle, cid, bases, scalars, n)
    ctx = ea.multi_scalar_mult_init(bases, curve.name)
    for opts in ({"window_bits": 2}, {"window_bits": 7}, {"window_bits": 11, "lane_entries": 4}, {"window_bits": 16, "lane_entries": 64},
                 {"lane_entries": 1000}, {"seg_entries": 4}, {"seg_entries": 5}, {"seg_entries": 64}, {"max_chunk": 1000}, {"max_chunk": 4096, "window_bits": 9}):
        for k in ("window_bits", "lane_entries", "seg_entries", "max_chunk"):
            ctx.set_option(k, opts.get(k, 0))
        assert ea.multi_scalar_mult(ctx, bases, scalars)[0] == exp, opts
    with pytest.raises(ea.MsmError):
        ctx.set_option("window_bits", 99)
    with pytest.raises(ea.MsmError):
        ctx.set_option("seg_entries", 2)      # a fan-in below 4 cannot shrink the fragment list
    with pytest.raises(ea.MsmError):
        ctx.set_option("nonsense", 1)
    ctx.close()


@pytest.mark.parametrize("cid,curve", CURVES)
def test_montgomery_form_scalars(ea, oracle, torch_cuda, cid, curve):
    """VariableBaseMSM::msm(bases, &[Fr]): scalars arrive as Fr values (a * 2^256 mod r) and are converted on the device
    (`into_bigint`, ARK ec/src/msm/variable_base/mod.rs:48-53); the result equals msm_bigint on the plain integers."""
    rng = random.Random(21 + cid)
    n = 3000
    ks = m.random_scalars(curve, n, rng)
    ks[0], ks[1], ks[2] = 0, 1, curve.r - 1
    bases = ea.generate_points(n, distinct=100, seed=4, curve=curve.name)
    plain = np.frombuffer(m.encode_scalars(ks), dtype=np.uint8).reshape(n, 32)
    mont = np.frombuffer(m.encode_scalars([(k << 256) % curve.r for k in ks]), dtype=np.uint8).reshape(n, 32)
    exp = oracle_msm_np(oracle, cid, bases, np.ascontiguousarray(plain), n)
    ctx = ea.multi_scalar_mult_init(bases, curve.name)
    assert ea.multi_scalar_mult(ctx, bases, plain)[0] == exp
    ctx.set_option("scalars_montgomery", 1)
    assert ea.multi_scalar_mult(ctx, bases, mont)[0] == exp
    ctx.set_option("scalars_montgomery", 0)
    assert ea.multi_scalar_mult(ctx, bases, plain)[0] == exp
    ctx.close()


def test_msm_chunks_streaming(ea, oracle, torch_cuda):
    """VariableBaseMSM::msm_chunks (ARK ec/src/msm/variable_base/mod.rs:165-199): Fr scalars, the last len(scalars) bases,
    `step` pairs at a time, partial sums added; any step gives the point msm_bigint gives on the aligned inputs."""
    cid, curve = 0, m.BLS12_377_G1
    rng = random.Random(77)
    nb, ns = 2700, 2500
    ks = m.random_scalars(curve, ns, rng)
    ks[0], ks[-1] = 0, curve.r - 1
    bases = ea.generate_points(nb, distinct=300, seed=8, curve=curve.name)
    plain = np.frombuffer(m.encode_scalars(ks), dtype=np.uint8).reshape(ns, 32)
    mont = np.frombuffer(m.encode_scalars([(k << 256) % curve.r for k in ks]), dtype=np.uint8).reshape(ns, 32)
    exp = oracle_msm_np(oracle, cid, np.ascontiguousarray(bases[nb - ns:]), np.ascontiguousarray(plain), ns)
    v = ea.VariableBaseMSM(curve.name)
    assert v.msm_chunks(bases, mont) == exp                       # one step covers everything
    assert v.msm_chunks(bases, mont, step=1000) == exp            # 1000 + 1000 + 500
    assert v.msm_chunks(bases.tobytes(), mont.tobytes(), step=999) == exp
    assert v.msm_chunks(bases, mont[:0]) == curve.encode_projective_normalized(None)
    with pytest.raises(ea.MsmError):
        v.msm_chunks(bases[:10], mont)                            # assert!(scalars_stream.len() <= bases_stream.len())


@pytest.mark.parametrize("cid,curve", CURVES)
def test_precomputed_tables(ea, oracle, golden, torch_cuda, cid, curve):
    """Row f1: a context with precomputed 2^(c w) P tables (all digits share one bucket set) returns the same bytes,
    for random inputs, for prefixes, with chunking, and for the low-order edge points (whose multiples hit infinity)."""
    n = 1 << 13
    bases = ea.generate_points(n, distinct=500, seed=12, curve=curve.name)
    bases[7, 96] = 1                      # a base flagged infinite
    scalars = rand_scalars_np(cid, 2 * n, 31)
    exp = [oracle_msm_np(oracle, cid, bases, np.ascontiguousarray(scalars[b * n:(b + 1) * n]), n) for b in range(2)]
    ctx = ea.MultiScalarMultContext(curve.name)
    ctx.set_option("precompute", 1)
    ctx.set_bases(bases)
    assert ctx.run(scalars) == exp
    t = ctx.last_timings()
    assert t["windows"] * n == t["entries"]
    assert ctx.run(np.ascontiguousarray(scalars[:1000]), npoints=1000)[0] == oracle_msm_np(
        oracle, cid, bases, np.ascontiguousarray(scalars[:1000]), 1000)
    ctx.set_option("max_chunk", 3000)
    assert ctx.run(scalars) == exp
    with pytest.raises(ea.MsmError):
        ctx.set_option("window_bits", 5)   # fixed by the tables
    ctx.close()
    for case in golden:
        if case["curve"] != curve.name:
            continue
        b, sc = bytes.fromhex(case["bases"]), bytes.fromhex(case["scalars"])
        ctx = ea.MultiScalarMultContext(curve.name)
        ctx.set_option("precompute", 1)
        ctx.set_option("window_bits", 6)
        ctx.set_bases(b)
        assert ctx.run(sc)[0].hex() == case["expected"], case["name"]
        ctx.close()


@pytest.mark.parametrize("curve_name,cid,rid", [("bls12_377_g1", 0, 0), ("bls12_381_g1", 1, 1), ("bls12_377_g2", 2, 0)])
def test_table_levels(ea, oracle, golden, torch_cuda, curve_name, cid, rid):
    """Row f1, the reference's own shape: k table levels 2^(c G j) P, windows g, g + G, ... sharing bucket set g (yrrid: k = 6, two
    bucket sets -- CMB PrecomputePoints.cu:10-39, MSM.cu:380-383).  Every k gives the oracle's bytes: random inputs with an
    infinity base, two batches, chunking over carried buckets, forced window sizes (incl. one whose windows do not divide by k),
    the low-order edge points of the golden vectors (their multiples hit infinity inside the tables)."""
    import ctypes

    stride = ea.affine_stride(curve_name)
    n = 6000
    bases = ea.generate_points(n, distinct=300, seed=21, curve=curve_name)
    bases[11, stride - 8] = 1
    scalars = rand_scalars_np(rid, 2 * n, 77)
    exp = []
    for b in range(2):
        out = ctypes.create_string_buffer(ea.projective_bytes(curve_name))
        sc = np.ascontiguousarray(scalars[b * n:(b + 1) * n])
        assert oracle.oracle_msm(cid, bases.ctypes.data, stride, sc.ctypes.data, n, out, 0) == 0
        exp.append(out.raw)
    for levels, wb in ((2, 0), (3, 0), (6, 0), (6, 7), (5, 11), (40, 9), (2, 13)):
        ctx = ea.MultiScalarMultContext(curve_name)
        ctx.set_option("precompute", 1)
        ctx.set_option("table_levels", levels)
        if wb:
            ctx.set_option("window_bits", wb)
        ctx.set_bases(bases)
        t_levels, c = ctx.query("table_levels"), ctx.query("table_window_bits")
        windows = -(-257 // c)
        sets = -(-windows // min(levels, windows))
        assert t_levels == -(-windows // sets), (levels, wb, t_levels, c)
        assert ctx.run(scalars) == exp, (levels, wb)
        assert ctx.last_timings()["tables"]
        ctx.set_option("max_chunk", 2500)
        assert ctx.run(scalars) == exp, (levels, wb, "chunked")
        ctx.close()
    for case in golden:
        if case["curve"] != curve_name:
            continue
        b, sc = bytes.fromhex(case["bases"]), bytes.fromhex(case["scalars"])
        ctx = ea.MultiScalarMultContext(curve_name)
        ctx.set_option("precompute", 1)
        ctx.set_option("table_levels", 3)
        ctx.set_option("window_bits", 6)
        ctx.set_bases(b)
        assert ctx.run(sc)[0].hex() == case["expected"], case["name"]
        ctx.close()


def test_plan_matches_context_for_table_levels(ea, torch_cuda):
    """mi355_msm_plan with table levels plans what build_tables builds (one shared helper since round 5): window bits and
    levels of the plan equal the context's "table_window_bits" / "table_levels", and its bucket sets equal what the run reports."""
    n = 1 << 14
    bases = ea.generate_points(n, distinct=200, seed=4)
    sc = rand_scalars_np(0, n, 5)
    for k in (0, 2, 3, 6):
        ctx = ea.MultiScalarMultContext("bls12_377_g1")
        ctx.set_option("precompute", 1)
        ctx.set_option("table_levels", k)
        ctx.set_bases(bases)
        pl = ea.plan(n, "bls12_377_g1", precompute=True, table_levels=k)
        assert pl["window_bits"] == ctx.query("table_window_bits"), k
        levels = ctx.query("table_levels")
        assert pl["bucket_windows"] == -(-pl["windows"] // levels), (k, pl, levels)
        ctx.run(sc)
        assert ctx.query("bucket_windows") == pl["bucket_windows"] and ctx.last_timings()["window_bits"] == pl["window_bits"]
        ctx.close()


@pytest.mark.parametrize("curve_name,cid,rid", [("bls12_377_g1", 0, 0), ("bls12_381_g1", 1, 1)])
def test_precompute_auto(ea, oracle, torch_cuda, curve_name, cid, rid):
    """ "precompute" = 2: the context chooses its table levels from the free device memory at set_bases (csrc/msm_engine.hip
    precompute_auto_levels; the reference builds its tables in the untimed init, CMB MSM.cu:380-383).  Small inputs get none; at
    2^24 pairs the choice follows the memory it is given (test hook "mem_limit"): everything -> a level per window, less -> 6 / 4 / 3
    levels, too little -> none; a table build that fails all the same (injected) leaves the context on the table-free path.  Every
    choice returns the bytes of the table-free run, which an oracle-checked prefix pins."""
    torch = torch_cuda
    small = ea.generate_points(3000, distinct=100, seed=2, curve=curve_name)
    ctx = ea.MultiScalarMultContext(curve_name)
    ctx.set_option("precompute", 2)
    ctx.set_bases(small)
    assert ctx.query("precompute") == 2 and ctx.query("table_levels") == 1 and ctx.query("table_window_bits") == 0
    sc = rand_scalars_np(rid, 3000, 6)
    assert ctx.run(sc)[0] == oracle_msm_np(oracle, cid, small, sc, 3000)
    ctx.close()

    n, distinct = 1 << 24, 1 << 12
    tile = ea.generate_points(distinct, distinct=distinct, seed=8, curve=curve_name)
    bases = torch.from_numpy(tile).cuda().repeat(n // distinct, 1).contiguous()
    scal = torch.from_numpy(rand_scalars_np(rid, n, 9)).cuda()
    plain = ea.MultiScalarMultContext(curve_name)
    plain.set_bases(bases)
    ref = plain.run(scal)[0]
    k = 1 << 13
    assert plain.run(scal[:k].contiguous(), npoints=k)[0] == oracle_msm_np(
        oracle, cid, np.ascontiguousarray(np.tile(tile, (k // distinct, 1))), scal[:k].cpu().numpy(), k)
    plain.close()
    seen = []
    ctx = ea.MultiScalarMultContext(curve_name)
    ctx.set_option("precompute", 2)
    # every branch is driven through the test hook "mem_limit" (an upper bound on what the context may count as free), never through
    # what happens to be free on the box: 100 GB pays for a level per window at this size, 40 / 24 GB for fewer, 3 GB for none
    for limit_gb in (100, 40, 24, 3):
        ctx.set_option("mem_limit", limit_gb << 30)
        ctx.set_bases(bases)
        levels = ctx.query("table_levels")
        seen.append(levels)
        ctx.set_option("mem_limit", 0)
        assert ctx.run(scal)[0] == ref, (limit_gb, levels)
        assert ctx.last_timings()["tables"] == (levels > 1)
    assert seen[0] >= 9 and seen[-1] == 1 and seen[0] >= seen[1] >= seen[2] >= seen[3], seen
    assert any(1 < lv <= 6 for lv in seen), seen
    # a build that fails although the estimate said it fits: back to no tables, not an error
    ctx.set_option("inject_alloc_failures", 1)
    ctx.set_bases(bases)
    assert ctx.query("table_levels") == 1
    assert ctx.run(scal)[0] == ref
    ctx.close()


@pytest.mark.parametrize("curve_name,cid,rid,npow,expect", [("bls12_377_g1", 0, 0, 18, 6), ("bls12_377_g1", 0, 0, 20, 6), ("bls12_377_g1", 0, 0, 21, 1),
                                                             ("bls12_381_g1", 1, 1, 19, 6), ("bls12_381_g1", 1, 1, 20, 1),
                                                             ("bls12_377_g2", 2, 0, 18, 6), ("bls12_377_g2", 2, 0, 22, 1)])
def test_precompute_auto_at_mid_sizes(ea, oracle, torch_cuda, curve_name, cid, rid, npow, expect):
    """ "precompute" = 2 between 2^18 pairs and the per-curve upper bound takes SIX table levels (round 6: 6 - 23 % there,
    profiles/r06_size_sweep_tables.txt; the reference's own shape is 6 levels, CMB PrecomputePoints.cu:10-39), none just above the
    bound; the bytes are those of the table-free context, which the oracle pins at 2^18 on G1."""
    torch = torch_cuda
    n = 1 << npow
    tile = ea.generate_points(1 << 12, distinct=1 << 12, seed=18, curve=curve_name)
    bases = torch.from_numpy(tile).cuda().repeat(n >> 12, 1).contiguous()
    scal = torch.from_numpy(rand_scalars_np(rid, n, 19)).cuda()
    plain = ea.MultiScalarMultContext(curve_name)
    plain.set_bases(bases)
    ref = plain.run(scal)[0]
    plain.close()
    if cid < 2 and npow == 18:
        assert ref == oracle_msm_np(oracle, cid, np.ascontiguousarray(np.tile(tile, (n >> 12, 1))), scal.cpu().numpy(), n)
    ctx = ea.MultiScalarMultContext(curve_name)
    ctx.set_option("precompute", 2)
    ctx.set_bases(bases)
    assert ctx.query("table_levels") == expect, (curve_name, npow, ctx.query("table_levels"))
    assert ctx.run(scal)[0] == ref
    assert ctx.last_timings()["tables"] == (expect > 1)
    ctx.close()


def test_prefix_run_and_errors(ea, oracle, torch_cuda):
    c = m.BLS12_377_G1
    n = 600
    bases = ea.generate_points(n, distinct=50, seed=1, curve=c.name)
    scalars = rand_scalars_np(0, n, 3)
    ctx = ea.multi_scalar_mult_init(bases, c.name)
    got = ctx.run(np.ascontiguousarray(scalars[:100]), npoints=100)[0]
    assert got == oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(scalars[:100]), 100)
    with pytest.raises(ea.MsmError):
        ctx.run(np.zeros((n + 1, 32), dtype=np.uint8), npoints=n + 1)   # more points than uploaded bases
    with pytest.raises(ValueError):
        ctx.run(np.zeros((n + 1, 32), dtype=np.uint8))                  # not a whole number of batches
    ctx.close()


def test_randomized_sizes_and_knobs(ea, oracle, torch_cuda):
    """Fuzz: random sizes, curves, scalar shapes and tuning knobs against the CPU oracle (seeded, 40 cases)."""
    rng = random.Random(20260929)
    for case in range(40):
        cid, curve = CURVES[rng.randrange(2)]
        n = rng.choice([1, 2, 3, 17, 63, 64, 65, 255, 1000, 4095, 4097, 20000])
        bases = ea.generate_points(n, distinct=rng.choice([1, 2, 7, 100, n]), seed=case, curve=curve.name)
        for _ in range(rng.randrange(3)):
            bases[rng.randrange(n), 96] = 1                       # sprinkle infinity bases
        kind = rng.randrange(4)
        sc = rand_scalars_np(cid, n, 1000 + case)
        if kind == 1:
            sc[:, 4:] = 0                                         # 32-bit scalars: most windows empty
        elif kind == 2:
            sc[rng.randrange(n)] = 0
            sc[::3] = sc[0]                                       # repeated scalar: hot buckets
        elif kind == 3:
            sc[:, :16] = 0                                        # only high limbs set
        ctx = ea.MultiScalarMultContext(curve.name)
        pre = rng.random() < 0.3
        if pre:
            ctx.set_option("precompute", 1)
        if rng.random() < 0.6:
            ctx.set_option("window_bits", rng.randrange(2, 17))
        ctx.set_bases(bases)
        if rng.random() < 0.5:
            ctx.set_option("lane_entries", rng.choice([1, 2, 4, 8, 33, 500]))
        if rng.random() < 0.4:
            ctx.set_option("seg_entries", rng.choice([4, 5, 9, 100]))
        if rng.random() < 0.3:
            ctx.set_option("max_chunk", rng.choice([97, 1000, 5000]))
        got = ctx.run(sc)[0]
        ctx.close()
        assert got == oracle_msm_np(oracle, cid, bases, sc, n), (case, curve.name, n, kind, pre)


# ---- G2 (BASELINE.json configs[4]): Fq2 coordinates through the same kernels ------------------------------------

def _oracle_g2(oracle, bases_np, scalars_np, n, cid=2):
    out = ctypes.create_string_buffer(288)
    assert oracle.oracle_msm(cid, bases_np.ctypes.data, 200, scalars_np.ctypes.data, n, out, 0) == 0
    return out.raw


# (engine / oracle curve id, model, which r bounds the synthetic scalars)
G2_CURVES = [pytest.param(2, m.BLS12_377_G2, 0, id="bls12_377_g2"), pytest.param(3, m.BLS12_381_G2, 1, id="bls12_381_g2")]


@pytest.mark.parametrize("cid,c,rid", G2_CURVES)
@pytest.mark.parametrize("npow", [8, 12, 16])
def test_g2_random_vs_oracle(ea, oracle, torch_cuda, npow, cid, c, rid):
    n = 1 << npow
    bases = ea.generate_points(n, distinct=min(n, 256), seed=npow, curve=c.name)
    assert bases.shape == (n, 200)
    scalars = rand_scalars_np(rid, 2 * n, seed=500 + npow)
    ctx = ea.multi_scalar_mult_init(torch_cuda.from_numpy(bases).cuda(), c.name)
    got = ea.multi_scalar_mult(ctx, None, torch_cuda.from_numpy(scalars).cuda())
    assert len(got) == 2 and len(got[0]) == 288
    for b in range(2):
        assert got[b] == _oracle_g2(oracle, bases, np.ascontiguousarray(scalars[b * n:(b + 1) * n]), n, cid), (npow, b)
    ctx.close()


@pytest.mark.parametrize("cid,c,rid", G2_CURVES)
def test_g2_reference_generator_has_order_r_on_the_gpu(ea, golden_constants, torch_cuda, cid, c, rid):
    """The reference's G2 generator literals (tests/golden/constants.json, from ARKC bls12_377/src/curves/g2.rs:61-78 and
    bls12_381/src/curves/g2.rs:74-91) through the HIP path: r * G2 = O, (r - 1) * G2 = -G2, and 2^15 copies of G2 with scalars
    summing to r also vanish."""
    k = golden_constants[c.name]
    G = c.generator()
    assert (G[0].c0, G[0].c1, G[1].c0, G[1].c1) == tuple(int(k[n]) for n in ("GX0", "GX1", "GY0", "GY1")) and c.on_curve(G)
    base = c.encode_affine_array([G])
    assert ea.msm(base, m.encode_scalars([c.r]), c.name) == c.encode_projective_normalized(None)
    assert ea.msm(base, m.encode_scalars([c.r - 1]), c.name) == c.encode_projective_normalized(c.neg(G))
    n = 1 << 15
    rng = random.Random(15)
    sc = [rng.randrange(c.r) for _ in range(n - 1)]
    sc.append((-sum(sc)) % c.r)
    assert ea.msm(base * n, m.encode_scalars(sc), c.name) == c.encode_projective_normalized(None)


@pytest.mark.parametrize("cid,c,rid", G2_CURVES)
def test_g2_cofactor_known_answers_on_the_gpu(ea, golden_constants, torch_cuda, cid, c, rid):
    """The reference's G2 cofactor literals through the HIP path (GPU twin of tests/test_oracle.py::test_g2_cofactor_known_answers):
    COFACTOR_INV * (COFACTOR * G2) is the literal generator again, and COFACTOR * Q for a point Q of E'(Fq2) OFF the order-r subgroup
    (solved from the curve equation) is killed by r.  ARKC bls12_377/src/curves/g2.rs:17-34, bls12_381/src/curves/g2.rs:22-40."""
    from test_oracle import _cofactor_chunks, _g2_point_off_the_subgroup

    k = golden_constants[c.name]
    h, hinv = int(k["COFACTOR"]), int(k["COFACTOR_INV"])
    G = c.generator()
    pts, sc = _cofactor_chunks(c, G, h)
    hG = c.mul(h, G)
    assert ea.msm(c.encode_affine_array(pts), m.encode_scalars(sc), c.name) == c.encode_projective_normalized(hG)
    assert ea.msm(c.encode_affine_array([hG]), m.encode_scalars([hinv]), c.name) == c.encode_projective_normalized(G)
    Q = _g2_point_off_the_subgroup(c, 1000)
    pts, sc = _cofactor_chunks(c, Q, h)
    hQ = c.mul(h, Q)
    assert ea.msm(c.encode_affine_array(pts), m.encode_scalars(sc), c.name) == c.encode_projective_normalized(hQ)
    assert ea.msm(c.encode_affine_array([hQ]), m.encode_scalars([c.r]), c.name) == c.encode_projective_normalized(None)
    # the same pairs many times over on one context (the throughput kernels, two lanes per point): 512 copies of the chunk set
    reps = 512
    ctx = ea.multi_scalar_mult_init(c.encode_affine_array(pts) * reps, c.name)
    got = ea.multi_scalar_mult(ctx, None, m.encode_scalars(sc) * reps)[0]
    ctx.close()
    assert got == c.encode_projective_normalized(c.mul(reps, hQ))


@pytest.mark.parametrize("cid,c,rid", G2_CURVES)
def test_g2_edge_cases_and_stateless(ea, oracle, torch_cuda, cid, c, rid):
    rng = random.Random(77)
    for n in (0, 1, 2, 31, 33, 300):
        pts = m.random_points(c, n, rng, max(1, n // 4)) if n else []
        sc = m.random_scalars(c, n, rng)
        if n > 4:
            sc[0], sc[1], pts[2] = 0, 1, None
        bases, scalars = c.encode_affine_array(pts), m.encode_scalars(sc)
        got = ea.msm(bases, scalars, c.name)
        assert got == c.encode_projective_normalized(c.msm_pippenger(pts, sc) if n > 40 else c.msm_naive(pts, sc)), n
    # one hot bucket, cancellation to infinity, shard-and-fold
    n = 2000
    bases = ea.generate_points(n, distinct=16, seed=2, curve=c.name)
    ctx = ea.multi_scalar_mult_init(bases, c.name)
    same = np.tile(rand_scalars_np(rid, 1, 3), (n, 1))
    assert ea.multi_scalar_mult(ctx, bases, same)[0] == _oracle_g2(oracle, bases, np.ascontiguousarray(same), n, cid)
    sc = rand_scalars_np(rid, n, 4)
    whole = ea.multi_scalar_mult(ctx, bases, sc)[0]
    lo = ctx.run(np.ascontiguousarray(sc[:n // 2]), npoints=n // 2)[0]
    ctx2 = ea.multi_scalar_mult_init(np.ascontiguousarray(bases[n // 2:]), c.name)
    hi = ctx2.run(np.ascontiguousarray(sc[n // 2:]))[0]
    assert ea.fold_partials([lo, hi], c.name) == whole == _oracle_g2(oracle, bases, sc, n, cid)
    ctx.close()
    ctx2.close()


@pytest.mark.parametrize("cid,c,rid", G2_CURVES)
@pytest.mark.parametrize("npow", [20, 24])
def test_g2_full_size_properties(ea, oracle, torch_cuda, npow, cid, c, rid):
    """G2 at 2^24 pairs (BASELINE.json configs[4] is the BLS12-377 one): linearity + prefix parity (the CPU oracle needs minutes at this size)."""
    torch = torch_cuda
    n = 1 << npow
    distinct = 1 << 12
    tile = ea.generate_points(distinct, distinct=distinct, seed=6, curve=c.name)
    bases = torch.from_numpy(tile).cuda().repeat(n // distinct, 1).contiguous()
    g = torch.Generator(device="cuda")
    g.manual_seed(npow)
    k1 = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    k1[:, 3] &= (1 << 59) - 1
    k2 = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    k2[:, 3] &= (1 << 59) - 1
    k2[:, :3] = 0            # k2 = t * 2^192: adding it touches only the top limb, no carries
    as_bytes = lambda t: t.view(torch.uint8).reshape(-1, 32)
    ksum = k1.clone()
    ksum[:, 3] += k2[:, 3]
    ctx = ea.MultiScalarMultContext(c.name)
    ctx.set_bases(bases)
    r1 = ctx.run(as_bytes(k1))[0]
    r2 = ctx.run(as_bytes(k2))[0]
    r12 = ctx.run(as_bytes(ksum))[0]
    assert ea.fold_partials([r1, r2], c.name) == r12
    sample = 1 << 13
    sc = as_bytes(k1[:sample].contiguous()).cpu().numpy()
    assert ctx.run(as_bytes(k1[:sample].contiguous()), npoints=sample)[0] == _oracle_g2(
        oracle, np.ascontiguousarray(np.tile(tile, (sample // distinct, 1))), sc, sample, cid)
    print("%s 2^%d timings: %s" % (c.name, npow, ctx.last_timings()))
    ctx.close()


# ---- full sizes: properties instead of the (too slow) oracle -------------------------------------------------

def _big_case(ea, torch_cuda, curve, cid, npow, seed):
    n = 1 << npow
    distinct = 1 << 15
    tile = ea.generate_points(distinct, distinct=distinct, seed=seed, curve=curve.name)
    bases = torch_cuda.from_numpy(tile).cuda().repeat(n // distinct, 1).contiguous()
    g = torch_cuda.Generator(device="cuda")
    g.manual_seed(seed)
    limbs = torch_cuda.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch_cuda.int64, device="cuda", generator=g)
    limbs[:, 3] &= (1 << 59) - 1   # < 2^251: sums of two stay below 2^252 < r, no carries out of 256 bits
    return tile, bases, limbs


@pytest.mark.parametrize("cid,curve,npow", [(0, m.BLS12_377_G1, 22), (0, m.BLS12_377_G1, 26), (1, m.BLS12_381_G1, 26)])
def test_full_size_properties(ea, oracle, torch_cuda, cid, curve, npow):
    torch = torch_cuda
    n = 1 << npow
    tile, bases, k1 = _big_case(ea, torch, curve, cid, npow, seed=40 + npow + cid)
    ctx = ea.MultiScalarMultContext(curve.name)
    ctx.set_bases(bases)
    as_bytes = lambda t: t.view(torch.uint8).reshape(-1, 32)
    # (1) all-ones checksum: sum of the bases = (n / 2^15) * sum of the 2^15 distinct points (oracle at 2^15)
    ones = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    ones[:, 0] = 1
    got_ones = ctx.run(as_bytes(ones))[0]
    tile_sum = oracle_msm_np(oracle, cid, tile, np.tile(np.array([1] + [0] * 31, dtype=np.uint8), (1 << 15, 1)), 1 << 15)
    mult = np.zeros((1, 32), dtype=np.uint8)
    mult[0, :8] = np.frombuffer((n >> 15).to_bytes(8, "little"), dtype=np.uint8)
    aff = tile_sum[:96] + b"\x00" * 8
    assert got_ones == oracle_msm(oracle, cid, aff, mult.tobytes(), 1)
    # (2) linearity: msm(k1) + msm(k2) == msm(k1 + k2)   (limb-wise int64 add with manual carries)
    g = torch.Generator(device="cuda")
    g.manual_seed(999)
    k2 = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    k2[:, 3] &= (1 << 59) - 1
    ksum = torch.empty_like(k1)
    carry = torch.zeros(n, dtype=torch.int64, device="cuda")
    for j in range(4):
        a, b = k1[:, j], k2[:, j]
        s = a + b
        c1 = ((a < 0) & (b < 0)) | (((a < 0) | (b < 0)) & (s >= 0))      # unsigned overflow of a + b
        s2 = s + carry
        c2 = (s < 0) & (s2 >= 0) & (carry > 0)                             # unsigned overflow of + carry
        ksum[:, j] = s2
        carry = (c1 | c2).to(torch.int64)
    r1 = ctx.run(as_bytes(k1))[0]
    r2 = ctx.run(as_bytes(k2))[0]
    r12 = ctx.run(as_bytes(ksum))[0]
    assert ea.fold_partials([r1, r2], curve.name) == r12
    # (3) shard-and-fold: the MSM over all pairs equals the fold of the MSMs over two disjoint halves
    half = n // 2
    lo = ctx.run(as_bytes(k1[:half].contiguous()), npoints=half)[0]
    ctx_hi = ea.MultiScalarMultContext(curve.name)
    ctx_hi.set_bases(bases[half:].contiguous())
    hi = ctx_hi.run(as_bytes(k1[half:].contiguous()))[0]
    assert ea.fold_partials([lo, hi], curve.name) == r1
    # (4) spot parity on a prefix the oracle can do
    sample = 1 << 16
    sc = as_bytes(k1[:sample].contiguous()).cpu().numpy()
    assert ctx.run(as_bytes(k1[:sample].contiguous()), npoints=sample)[0] == oracle_msm_np(
        oracle, cid, np.ascontiguousarray(np.tile(tile, (sample >> 15, 1))), sc, sample)
    # (5) idempotence / determinism
    assert ctx.run(as_bytes(k1))[0] == r1
    # (6) internal chunking at scale: the same MSM through chunks of 2^(npow-1) - 12345 pairs (three chunks, large base offsets)
    ctx.set_option("max_chunk", (n // 2) - 12345)
    assert ctx.run(as_bytes(k1))[0] == r1
    ctx.set_option("max_chunk", 0)
    # (7) a context with precomputed tables gives the same point
    if npow <= 24 or cid == 0:
        ctx_pre = ea.MultiScalarMultContext(curve.name)
        ctx_pre.set_option("precompute", 1)
        ctx_pre.set_bases(bases)
        assert ctx_pre.run(as_bytes(k1))[0] == r1
        ctx_pre.close()
    # (8) the harness's hand-over: scalars in HOST memory, two batches -- the first batch's copy is split 1/26 + 3/26 + 9/26 + the rest
    #     (the Edwards kernels accumulate every piece onto ONE bucket array; CMB MSM.cu:419-434 splits 1/4 + 3/4), the second batch is
    #     copied while the first computes
    if npow >= 23 and cid == 0:
        host = torch.cat([as_bytes(k1), as_bytes(k2)]).cpu().numpy()
        assert ctx.run(host) == [r1, r2]
        assert ctx.last_timings()["launches"] == 5          # 4 chunks for batch 0, 1 for batch 1
    # (9) BLS12-377: the twisted-Edwards path (default) and the XYZZ path agree at full size
    if cid == 0:
        assert ctx.query("twisted_edwards") == 1 and ctx.query("twisted_edwards_fallbacks") == 0
        ctx_sw = ea.MultiScalarMultContext(curve.name)
        ctx_sw.set_option("twisted_edwards", 0)
        ctx_sw.set_bases(bases)
        assert ctx_sw.query("twisted_edwards") == 0 and ctx_sw.run(as_bytes(k1))[0] == r1
        ctx_sw.close()
    ctx.close()
    ctx_hi.close()
