"""GPU: the engine's back-off paths, forced.  (1) chunks are sized to the device memory that is free (the reference plans its
allocations before it runs, ML msm.cu:453-466): the "mem_limit" test hook shrinks the budget; (2) an allocation that fails all
the same halves the chunk and retries ("inject_alloc_failures"); (3) a base set that trips the incomplete twisted-Edwards
law twice in a row is demoted to XYZZ instead of paying for both paths on every call.  Results never change."""
import random

import numpy as np
import pytest

import pymodel as m
import te_model as te
from conftest import oracle_msm_np

pytestmark = pytest.mark.gpu

C = m.BLS12_377_G1


def _scalars(n, seed):
    rng = np.random.default_rng(seed)
    limbs = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] %= np.uint64(0x12ab655e9a2ca556)
    return limbs.view(np.uint8).reshape(n, 32)


@pytest.mark.parametrize("curve,cid", [("bls12_377_g1", 0), ("bls12_381_g1", 1)])
def test_chunks_shrink_to_the_memory_budget(ea, oracle, curve, cid):
    n = 1 << 17
    bases = ea.generate_points(n, distinct=1 << 10, seed=6, curve=curve)
    sc = _scalars(n, 2)
    sc[:, 31] &= 0x0F
    exp = oracle_msm_np(oracle, cid, bases, sc, n)
    ctx = ea.multi_scalar_mult_init(bases, curve)
    assert ctx.run(sc)[0] == exp and ctx.last_timings()["launches"] == 1
    need = ea.plan(n, curve)["work_bytes"]
    ctx.set_option("mem_limit", need // 3)          # a third of what one chunk of n needs: four chunks of n/4
    assert ctx.run(sc)[0] == exp
    assert ctx.last_timings()["launches"] >= 3
    ctx.set_option("mem_limit", 0)
    assert ctx.run(sc)[0] == exp and ctx.last_timings()["launches"] == 1
    ctx.close()


def test_allocation_failure_halves_the_chunk_and_retries(ea, oracle):
    ea.trim()        # (a parked stateless context would be reclaimed first and the same chunk retried: tests/test_gpu_stateless.py)
    n = 60000
    bases = ea.generate_points(n, distinct=777, seed=9)
    sc = _scalars(2 * n, 4)
    exp = [oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(sc[b * n:(b + 1) * n]), n) for b in range(2)]
    ctx = ea.multi_scalar_mult_init(bases, "bls12_377_g1")
    ctx.set_option("inject_alloc_failures", 1)       # the first buffer reservation of the next run fails
    assert ctx.run(sc) == exp
    assert ctx.query("oom_backoffs") == 1 and 0 < ctx.query("chunk_cap") <= (n + 1) // 2
    assert ctx.last_timings()["launches"] >= 4       # two batches, each in (at least) two chunks now
    # the cap lasted for that run only: one transient failure does not tax every later MSM
    assert ctx.run(np.ascontiguousarray(sc[:n]))[0] == exp[0] and ctx.last_timings()["launches"] == 1
    assert ctx.query("chunk_cap") == 0 and ctx.query("oom_backoffs") == 1
    ctx.close()


def test_allocation_failure_backoff_in_every_shard_of_a_sharded_context(ea, oracle):
    """The injection counter is a field of each shard's context (it used to be a thread-local of the CALLING thread, which the
    shards' worker threads never saw)."""
    ea.trim()
    n = 50000
    bases = ea.generate_points(n, distinct=333, seed=10)
    sc = _scalars(n, 5)
    exp = oracle_msm_np(oracle, 0, bases, sc, n)
    ctx = ea.multi_scalar_mult_init(bases, "bls12_377_g1", devices=[0, 0, 0])
    assert ctx.run(sc)[0] == exp and ctx.query("oom_backoffs") == 0
    ctx.set_option("inject_alloc_failures", 1)
    assert ctx.run(sc)[0] == exp
    assert ctx.query("oom_backoffs") == 3           # one per shard
    assert ctx.run(sc)[0] == exp and ctx.query("oom_backoffs") == 3
    ctx.close()


def _failing_pair():
    rng = random.Random(9)
    for P in m.random_points(C, 8, rng):
        for E in te.exceptional_points():
            R = C.add(P, E)
            a, b = te.sw_to_te(R), te.sw_to_te(P)
            if a is None or b is None:
                continue
            if te.te_add(a, b) is None:
                return R, P
            if te.te_add(a, te.te_neg(b)) is None:
                return R, C.neg(P)
    raise AssertionError("no failing pair found")


def test_two_fallbacks_in_a_row_demote_the_context(ea, oracle):
    R, Q = _failing_pair()
    rng = random.Random(4)
    n = 3000
    pts = m.random_points(C, 50, rng)
    seq = [pts[i % 50] for i in range(n)]
    seq[100], seq[2000] = R, Q
    bases = np.frombuffer(C.encode_affine_array(seq), dtype=np.uint8).reshape(n, 104).copy()
    bad = np.zeros((n, 32), dtype=np.uint8)
    bad[100] = bad[2000] = _scalars(1, 5)[0]
    good = _scalars(n, 6)
    good[100] = 0
    ctx = ea.multi_scalar_mult_init(bases, "bls12_377_g1")
    assert ctx.query("twisted_edwards") == 1
    exp_bad = oracle_msm_np(oracle, 0, bases, bad, n)
    assert ctx.run(bad)[0] == exp_bad and ctx.query("twisted_edwards_fallbacks") == 1
    # a clean run in between resets the streak
    assert ctx.run(good)[0] == oracle_msm_np(oracle, 0, bases, good, n)
    assert ctx.run(bad)[0] == exp_bad and ctx.query("twisted_edwards") == 1 and ctx.query("twisted_edwards_demotions") == 0
    # two in a row: demoted, no further fallbacks are paid for
    assert ctx.run(bad)[0] == exp_bad
    assert ctx.query("twisted_edwards") == 0 and ctx.query("twisted_edwards_demotions") == 1
    before = ctx.query("twisted_edwards_fallbacks")
    assert ctx.run(bad)[0] == exp_bad and ctx.query("twisted_edwards_fallbacks") == before
    assert ctx.last_timings()["twisted_edwards"] is False
    # a new base set gets a new chance
    ctx.set_bases(ea.generate_points(n, distinct=100, seed=3))
    assert ctx.query("twisted_edwards") == 1
    ctx.close()


@pytest.mark.parametrize("curve,cid", [("bls12_377_g1", 0), ("bls12_381_g1", 1), ("bls12_377_g2", 2), ("bls12_381_g2", 3)])
def test_scan_and_chunked_bucket_reductions_agree(ea, oracle, curve, cid):
    """Small windows reduce their buckets by a parallel scan (one addition per thread and step), large ones by chunked running
    sums; "reduce_scan" forces either.  Same bytes, also with empty buckets, a single bucket per window and sparse windows."""
    import ctypes

    stride = ea.affine_stride(curve)
    # window_bits <= 13: the scan runs on the buckets directly; 14 and 17: one chunked level first, then scan + join + tree
    for n, wb in ((1, 2), (37, 2), (500, 5), (3000, 9), (20000, 13), (4097, 0), (20000, 14), (5000, 17)):
        if cid >= 2 and n > 5000:
            continue
        bases = ea.generate_points(n, distinct=max(1, n // 7), seed=n, curve=curve)
        sc = _scalars(n, n + wb)
        sc[:, 31] &= 0x0F
        if n > 10:
            sc[3] = 0
            sc[4, 1:] = 0          # a small scalar: upper windows stay empty
        exp = ctypes.create_string_buffer(ea.projective_bytes(curve))
        assert oracle.oracle_msm(cid, bases.ctypes.data, stride, sc.ctypes.data, n, exp, 0) == 0
        ctx = ea.MultiScalarMultContext(curve)
        if wb:
            ctx.set_option("window_bits", wb)
        ctx.set_bases(bases)
        for mode in (0, 1, -1):
            ctx.set_option("reduce_scan", mode)
            assert ctx.run(sc)[0] == exp.raw, (curve, n, wb, mode)
        ctx.close()


@pytest.mark.parametrize("precompute", [0, 1])
def test_large_forced_windows_on_small_inputs(ea, oracle, precompute):
    """window_bits up to 24 on inputs of a few thousand pairs: the grouping then has far more bucket bits than entries to
    tell apart (two generic passes, level 1 must leave <= 15 bits for the key word's low half) -- results must not care."""
    n = 3000
    bases = ea.generate_points(n, distinct=211, seed=12)
    sc = _scalars(n, 77)
    exp = oracle_msm_np(oracle, 0, bases, sc, n)
    for wb in (17, 20, 21, 22, 24):
        ctx = ea.MultiScalarMultContext("bls12_377_g1")
        ctx.set_option("window_bits", wb)
        ctx.set_option("precompute", precompute)
        ctx.set_bases(bases)
        assert ctx.run(sc)[0] == exp, (wb, precompute)
        assert ctx.last_timings()["window_bits"] == wb
        ctx.close()


@pytest.mark.parametrize("curve,cid,te", [("bls12_377_g1", 0, 1), ("bls12_377_g1", 0, 0), ("bls12_381_g1", 1, 0), ("bls12_377_g2", 2, 0), ("bls12_381_g2", 3, 0)])
def test_quad_and_single_lane_additions_agree(ea, oracle, curve, cid, te):
    """G1 contexts run the fragment merge and the scan reduction with FOUR LANES PER ADDITION below "quad_limit" additions per
    launch (te.hpp te_add_quad, curve.hpp xyzz_add_quad) and one lane per addition above it.  Both forms, and a limit that
    splits the levels of one MSM between them, give the oracle's bytes -- with skewed scalars (long runs; equal bases under
    equal scalars: the XYZZ law's doubling case), empty buckets, cancelling pairs and a ragged size."""
    import ctypes

    stride = ea.affine_stride(curve)
    if True:
        for n, wb, fan in ((1, 0, 0), (97, 0, 0), (1000, 7, 4), (4099, 0, 5), (30000, 11, 0), (70001, 0, 8)):
            if cid >= 2 and n > 5000:
                continue
            bases = ea.generate_points(n, distinct=max(1, n // 5), seed=n + 1, curve=curve)
            sc = _scalars(n, 3 * n + wb)
            sc[:, 31] &= 0x0F
            if n > 10:
                sc[: n // 3] = sc[0]       # a third of the scalars equal: long runs in every window
                sc[n // 2, 1:] = 0         # a small scalar: upper windows stay empty
            exp = ctypes.create_string_buffer(ea.projective_bytes(curve))
            assert oracle.oracle_msm(cid, bases.ctypes.data, stride, sc.ctypes.data, n, exp, 0) == 0
            ctx = ea.MultiScalarMultContext(curve)
            if wb:
                ctx.set_option("window_bits", wb)
            if fan:
                ctx.set_option("seg_entries", fan)
            if cid == 0:
                ctx.set_option("twisted_edwards", te)
            ctx.set_bases(bases)
            for limit in (0, 1 << 18, 300, 1 << 24):
                ctx.set_option("quad_limit", limit)
                assert ctx.run(sc)[0] == exp.raw, (curve, te, n, wb, fan, limit)
                assert ctx.query("twisted_edwards") == te
            ctx.close()       # "quad_limit" is a field of the context: nothing to restore


@pytest.mark.parametrize("curve,cid", [("bls12_377_g1", 0), ("bls12_381_g1", 1)])
def test_quad_xyzz_addition_special_cases(ea, oracle, curve, cid):
    """The XYZZ quad addition decides infinity operands, the doubling and the cancellation quad-uniformly: P and P in one bucket
    from different lanes (doubling in the merge), P and -P (cancellation to infinity), buckets that stay empty, and window sums
    that are equal or opposite (the scan / tree steps add them)."""
    import ctypes

    C = m.BLS12_377_G1 if cid == 0 else m.BLS12_381_G1
    rng = random.Random(17 + cid)
    stride = ea.affine_stride(curve)
    P, Q = C.mul(5, C.generator()), C.mul(7, C.generator())
    cases = {
        "same point many times, equal scalars": ([P] * 64, [rng.randrange(1, C.r)] * 64),
        "pairs that cancel": ([P, C.neg(P)] * 20 + [Q], [12345] * 40 + [3]),
        "two blocks whose sums are equal": ([P] * 16 + [Q] * 16, [3 << 8] * 16 + [(3 << 8) + 0] * 16),
        "everything cancels": ([P, C.neg(P)] * 32, [(1 << 200) + 77] * 64),
    }
    for name, (pts, ks) in cases.items():
        n = len(pts)
        bases = np.frombuffer(C.encode_affine_array(pts), dtype=np.uint8).reshape(n, stride).copy()
        sc = np.frombuffer(b"".join(k.to_bytes(32, "little") for k in ks), dtype=np.uint8).reshape(n, 32).copy()
        exp = ctypes.create_string_buffer(ea.projective_bytes(curve))
        assert oracle.oracle_msm(cid, bases.ctypes.data, stride, sc.ctypes.data, n, exp, 0) == 0
        for wb, lanes in ((0, 0), (4, 4), (9, 4)):
            ctx = ea.MultiScalarMultContext(curve)
            if cid == 0:
                ctx.set_option("twisted_edwards", 0)
            if wb:
                ctx.set_option("window_bits", wb)
            if lanes:
                ctx.set_option("lane_entries", lanes)
            ctx.set_bases(bases)
            assert ctx.run(sc)[0] == exp.raw, (curve, name, wb, lanes)
            ctx.close()
