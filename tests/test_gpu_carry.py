"""GPU: carried buckets.  A batch that runs as several chunks (max_chunk, a memory budget, the first piece of a host-scalar batch,
the slices of the stateless call) keeps ONE bucket array: every chunk accumulates with the window size of the whole batch, chunk 0's
buckets become the batch's, later chunks are added to them (k_bucket_merge; the twisted-Edwards kernels since round 6 accumulate straight onto the stored buckets instead: SegOutT::carry_in) and only the last chunk reduces.  The reference's
counterpart is the single bucket set its batches of points feed (CMB MSM.cu:437-505 accumulates all its point groups before
ReduceBuckets runs once).  Results are bit-exact against the CPU oracle and against the per-chunk-reduce path (option carry = 0)."""
import random

import numpy as np
import pytest

import pymodel as m
import te_model as te
from conftest import oracle_msm_np

pytestmark = pytest.mark.gpu

R_TOP = {0: 0x12ab655e9a2ca556, 1: 0x73eda753299d7d48, 2: 0x12ab655e9a2ca556, 3: 0x73eda753299d7d48}
NAMES = {0: "bls12_377_g1", 1: "bls12_381_g1", 2: "bls12_377_g2", 3: "bls12_381_g2"}
STRIDE = {0: 104, 1: 104, 2: 200, 3: 200}


def _scalars(cid, n, seed):
    rng = np.random.default_rng(seed)
    limbs = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] %= np.uint64(R_TOP[cid])
    return limbs.view(np.uint8).reshape(n, 32)


def _oracle(oracle, cid, bases, sc, n):
    out = np.zeros(288 if cid >= 2 else 144, dtype=np.uint8)
    assert oracle.oracle_msm(cid, bases.ctypes.data, STRIDE[cid], np.ascontiguousarray(sc).ctypes.data, n, out.ctypes.data, 0) == 0
    return out.tobytes()


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_chunked_batches_carry_their_buckets(ea, oracle, cid):
    n = 20000 if cid < 2 else 6000
    bases = ea.generate_points(n, distinct=211, seed=13 + cid, curve=NAMES[cid])
    sc = _scalars(cid, 2 * n, 3)
    sc[7] = 0
    sc[n - 1] = sc[0]          # equal scalars in the first and the last chunk: their bases meet in the merge
    exp = [_oracle(oracle, cid, bases, sc[b * n:(b + 1) * n], n) for b in range(2)]
    ctx = ea.multi_scalar_mult_init(bases, NAMES[cid])
    whole = ctx.run(sc)
    assert whole == exp and ctx.last_timings()["launches"] == 2
    c_whole = ctx.last_timings()["window_bits"]
    for chunk in (n // 2 + 1, n // 7, 1025):
        ctx.set_option("max_chunk", chunk)
        for carry in (1, 0):
            ctx.set_option("carry", carry)
            assert ctx.run(sc) == exp, (chunk, carry)
            t = ctx.last_timings()
            assert t["launches"] == 2 * -(-n // chunk)
            if carry:
                assert t["window_bits"] == c_whole      # every chunk ran with the window size of the whole batch
    # a forced window size, host and device scalars, the Fr-Montgomery entry
    ctx.set_option("carry", 1)
    ctx.set_option("max_chunk", 3000)
    for c in (5, 13):
        ctx.set_option("window_bits", c)
        assert ctx.run(sc) == exp, c
    ctx.set_option("window_bits", 0)
    import torch

    assert ctx.run(torch.from_numpy(sc).cuda()) == exp
    ctx.close()


def test_carried_tables_and_fold(ea, oracle):
    n = 9000
    bases = ea.generate_points(n, distinct=100, seed=5)
    sc = _scalars(0, n, 8)
    exp = _oracle(oracle, 0, bases, sc, n)
    ctx = ea.MultiScalarMultContext("bls12_377_g1")
    ctx.set_option("precompute", 1)
    ctx.set_bases(bases)
    ctx.set_option("max_chunk", 2000)
    ctx.set_option("assume_subgroup", 1)
    assert ctx.run(sc)[0] == exp and ctx.last_timings()["tables"] and ctx.last_timings()["launches"] == 5
    ctx.close()


def test_allocation_failure_in_the_middle_of_a_carried_batch(ea, oracle):
    """The totals of the chunks already merged survive the back-off (release_work_buffers keeps carry_buckets)."""
    ea.trim()        # nothing parked: the failure below must take the back-off path, not the reclaim-and-retry one
    n = 40000
    bases = ea.generate_points(n, distinct=500, seed=2)
    sc = _scalars(0, n, 6)
    exp = _oracle(oracle, 0, bases, sc, n)
    ctx = ea.multi_scalar_mult_init(bases, "bls12_377_g1")
    ctx.set_option("max_chunk", 10000)
    ctx.set_option("inject_alloc_failures", -3)      # the third reservation of the next run: chunk 2 of 4
    assert ctx.run(sc)[0] == exp
    assert ctx.query("oom_backoffs") == 1 and ctx.last_timings()["launches"] >= 6
    ctx.set_option("inject_alloc_failures", -1)      # the first chunk
    assert ctx.run(sc)[0] == exp and ctx.query("oom_backoffs") == 2
    ctx.close()


def _failing_pair():
    C = m.BLS12_377_G1
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


def test_twisted_edwards_failure_in_a_carried_batch_repeats_the_batch(ea, oracle):
    """Two bases whose Edwards images add to a vanishing denominator, in DIFFERENT chunks: they meet in k_bucket_merge.  The batch
    is repeated on the XYZZ path (its carried buckets were Edwards sums); twice in a row demotes the context."""
    C = m.BLS12_377_G1
    R, Q = _failing_pair()
    rng = random.Random(4)
    n = 3000
    pts = m.random_points(C, 50, rng)
    seq = [pts[i % 50] for i in range(n)]
    seq[100], seq[2000] = R, Q
    bases = np.frombuffer(C.encode_affine_array(seq), dtype=np.uint8).reshape(n, 104).copy()
    bad = np.zeros((n, 32), dtype=np.uint8)
    bad[100] = bad[2000] = _scalars(0, 1, 5)[0]
    good = _scalars(0, n, 6)
    good[100] = 0
    ctx = ea.multi_scalar_mult_init(bases, "bls12_377_g1")
    ctx.set_option("max_chunk", 1000)
    assert ctx.query("twisted_edwards") == 1
    exp_bad = _oracle(oracle, 0, bases, bad, n)
    assert ctx.run(bad)[0] == exp_bad and ctx.query("twisted_edwards_fallbacks") == 1
    assert ctx.last_timings()["launches"] == 6 and not ctx.last_timings()["twisted_edwards"]      # three Edwards chunks, three XYZZ ones
    assert ctx.run(good)[0] == _oracle(oracle, 0, bases, good, n) and ctx.last_timings()["twisted_edwards"]
    assert ctx.run(bad)[0] == exp_bad and ctx.query("twisted_edwards") == 1 and ctx.query("twisted_edwards_demotions") == 0
    assert ctx.run(bad)[0] == exp_bad
    assert ctx.query("twisted_edwards") == 0 and ctx.query("twisted_edwards_demotions") == 1
    ctx.close()


@pytest.mark.parametrize("cid", [0, 1])
def test_host_scalar_batches_split_their_first_piece(ea, cid):
    """2^23 pairs from host memory: the first piece of batch 0 is computed while the rest crosses PCIe, as chunks of ONE carried batch.
    Same bytes as with device-resident scalars (a single chunk), for several piece sizes and without carrying."""
    import torch

    n = 1 << 23
    bases = ea.generate_points(n, distinct=1 << 12, seed=3, curve=NAMES[cid])
    sc = _scalars(cid, 2 * n, 11)
    ctx = ea.MultiScalarMultContext(NAMES[cid])
    ctx.set_bases(torch.from_numpy(bases).cuda())
    ref = ctx.run(torch.from_numpy(sc).cuda())
    assert ctx.last_timings()["launches"] == 2
    # (first piece 1/div of the batch, every further piece three times its predecessor: 13 -> 1/13 + 3/13 + 9/13)
    # default divisor: 13 with a merge pass per piece (XYZZ), 26 when the pieces accumulate onto the stored buckets (the Edwards kernels of
    # BLS12-377 G1: 1/26 + 3/26 + 9/26 + the rest)
    for div, carry, pieces in ((0, 1, 4 if cid == 0 else 3), (13, 1, 3), (4, 1, 2), (40, 1, 4), (0, 0, 2), (16, 0, 3)):
        ctx.set_option("first_piece_div", div)
        ctx.set_option("carry", carry)
        assert ctx.run(sc) == ref, (div, carry)
        assert ctx.last_timings()["launches"] == pieces + 1, (div, carry)
    ctx.close()


def test_golden_vectors_in_carried_chunks(ea, golden):
    """Every golden case (incl. the FPGA harness's edge fixtures: P, -P, the 2-torsion point T twice, special points at chunk
    boundaries -- P1B hardcaml msm_unit_tests.rs:21-242) as chunks of 5 and of 16 pairs over carried buckets, and with per-chunk
    reductions: the special points meet at OUR chunk boundaries and in the bucket merge, on all three curves."""
    for case in golden:
        bases, scalars = bytes.fromhex(case["bases"]), bytes.fromhex(case["scalars"])
        ctx = ea.multi_scalar_mult_init(bases, case["curve"])
        for chunk in (5, 16):
            ctx.set_option("max_chunk", chunk)
            for carry in (1, 0):
                ctx.set_option("carry", carry)
                assert ea.multi_scalar_mult(ctx, bases, scalars)[0].hex() == case["expected"], (case["curve"], case["name"], chunk, carry)
        ctx.close()


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_first_reduce_level_with_chunks_that_fill_the_simds(ea, oracle, cid):
    """Large windows (>= 2^17 buckets, fewer than 16 windows): the first bucket-reduce level is cut into a non-power-of-two number of
    chunks (one wave per SIMD), X_t is scaled by a non-power-of-two L and the scan runs on padded rows -- against the oracle, with the
    options that switch the re-cut off (explicit chunk sizes) and the recursive scheme giving the same bytes."""
    n = 30000 if cid != 2 else 8000
    bases = ea.generate_points(n, distinct=300, seed=40 + cid, curve=NAMES[cid])
    sc = _scalars(cid, n, 12)
    sc[5] = 0
    exp = _oracle(oracle, cid, bases, sc, n)
    ctx = ea.multi_scalar_mult_init(bases, NAMES[cid])
    for c in (18, 19, 20, 22, 23):
        ctx.set_option("window_bits", c)
        assert ctx.run(sc)[0] == exp, c
        if c in (18, 20):
            ctx.set_option("reduce_log_chunk0", 7)      # explicit power-of-two chunks
            assert ctx.run(sc)[0] == exp, (c, "pow2")
            ctx.set_option("reduce_log_chunk0", 0)
            ctx.set_option("reduce_scan", 0)            # recursive chunked scheme only
            assert ctx.run(sc)[0] == exp, (c, "recursive")
            ctx.set_option("reduce_scan", -1)
    ctx.close()
