"""GPU: the G2 throughput kernels with every Fp2 value spread over TWO LANES (csrc/fp2pair.hpp, option "g2_paired") against the
CPU oracle, the golden vectors and the one-lane-per-point kernels.

Fp2 semantics: ARK ff/src/fields/models/quadratic_extension.rs:641-652 (the product), :273 (square); the group law is the same
curve.hpp template as every other path (SPK ec/xyzz_t.hpp:97-249).  Both forms read and write the SAME records, so every bit of
the mask is also tested on its own (a paired kernel feeding a one-lane kernel and vice versa), with `quad_limit` = 0 so that the
merge and scan launches of a small input take the throughput kernels at all."""
import json
import os
import random

import numpy as np
import pytest

import pymodel as m
from conftest import ROOT
from test_gpu_parity import G2_CURVES, _oracle_g2, rand_scalars_np

pytestmark = pytest.mark.gpu

ALL = 31   # accumulate | first reduce level | fragment merge | scan steps | bucket merge


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


@pytest.mark.parametrize("cid,c,rid", G2_CURVES)
@pytest.mark.parametrize("n", [1, 2, 63, 300, 4097, 1 << 14])
def test_paired_equals_oracle(ea, oracle, torch_cuda, cid, c, rid, n):
    bases = ea.generate_points(n, distinct=min(n, 128), seed=n, curve=c.name)
    scalars = rand_scalars_np(rid, n, seed=900 + n)
    if n > 4:
        scalars[0, :] = 0            # a zero scalar, a unit scalar, a base at infinity
        scalars[1, :] = 0
        scalars[1, 0] = 1
        bases[2, 192] = 1
    exp = _oracle_g2(oracle, bases, scalars, n, cid)
    ctx = ea.multi_scalar_mult_init(bases, c.name)
    ctx.set_option("quad_limit", 0)
    for mask in (0, ALL, 1, 2, 4, 8):
        ctx.set_option("g2_paired", mask)
        assert ctx.run(scalars)[0] == exp, (n, mask)
    ctx.close()


@pytest.mark.parametrize("cid,c,rid", G2_CURVES)
def test_paired_special_cases(ea, oracle, torch_cuda, cid, c, rid):
    """What the walk's rare branches see: one hot bucket (every scalar equal: long runs, lane-boundary fragments), P and -P in the
    same bucket (cancellation to infinity), repeated bases (the doubling branch re-reads the base from memory by halves), and a
    chunked batch over carried buckets (k_bucket_merge)."""
    n = 3000
    bases = ea.generate_points(n, distinct=8, seed=5, curve=c.name)
    ctx = ea.multi_scalar_mult_init(bases, c.name)
    ctx.set_option("quad_limit", 0)
    ctx.set_option("g2_paired", ALL)
    same = np.tile(rand_scalars_np(rid, 1, 3), (n, 1))
    assert ctx.run(same)[0] == _oracle_g2(oracle, bases, np.ascontiguousarray(same), n, cid)
    sc = rand_scalars_np(rid, n, 4)
    exp = _oracle_g2(oracle, bases, sc, n, cid)
    assert ctx.run(sc)[0] == exp
    ctx.set_option("max_chunk", n // 3 + 1)
    assert ctx.run(sc)[0] == exp
    ctx.set_option("max_chunk", 0)
    ctx.set_option("window_bits", 4)          # 8 buckets per window: P and -P meet, runs of hundreds
    assert ctx.run(sc)[0] == exp
    ctx.close()
    # P - P = O, and 2 * P through equal entries in one bucket
    P, Q = m.random_points(c, 2, random.Random(9))
    pts = [P, c.neg(P), Q, Q]
    b = c.encode_affine_array(pts)
    ctx = ea.multi_scalar_mult_init(b, c.name)
    ctx.set_option("g2_paired", ALL)
    ctx.set_option("quad_limit", 0)
    got = ctx.run(m.encode_scalars([7, 7, 5, 5]))[0]
    assert got == c.encode_projective_normalized(c.msm_naive(pts, [7, 7, 5, 5]))
    ctx.close()


def test_paired_golden_vectors(ea, golden, torch_cuda):
    for case in golden:
        if not case["curve"].endswith("g2"):
            continue
        bases, scalars = bytes.fromhex(case["bases"]), bytes.fromhex(case["scalars"])
        ctx = ea.multi_scalar_mult_init(bases, case["curve"])
        ctx.set_option("g2_paired", ALL)
        ctx.set_option("quad_limit", 0)
        assert ea.multi_scalar_mult(ctx, bases, scalars)[0].hex() == case["expected"], f'{case["curve"]}/{case["name"]}'
        ctx.close()
    from test_oracle import _expand_large

    for case in json.load(open(os.path.join(ROOT, "tests", "golden", "msm_vectors_large.json")))["cases"]:
        if not case["curve"].endswith("g2"):
            continue
        bases, scalars = _expand_large(case)
        ctx = ea.multi_scalar_mult_init(bases, case["curve"])
        ctx.set_option("g2_paired", ALL)
        assert ctx.run(scalars)[0].hex() == case["expected"], (case["curve"], case["n"])
        ctx.close()


@pytest.mark.parametrize("cid,c,rid", G2_CURVES)
def test_paired_full_size_equals_one_lane_form(ea, oracle, torch_cuda, cid, c, rid):
    """2^22 pairs (the window size and lane geometry of the 2^24 workload's regime): both forms must return the same bytes, and an
    oracle-checked prefix pins them."""
    torch = torch_cuda
    n = 1 << 22
    distinct = 1 << 12
    tile = ea.generate_points(distinct, distinct=distinct, seed=6, curve=c.name)
    bases = torch.from_numpy(tile).cuda().repeat(n // distinct, 1).contiguous()
    sc = torch.from_numpy(rand_scalars_np(rid, n, 22)).cuda()
    ctx = ea.MultiScalarMultContext(c.name)
    ctx.set_bases(bases)
    ctx.set_option("g2_paired", 0)
    one = ctx.run(sc)[0]
    ctx.set_option("g2_paired", ALL)
    two = ctx.run(sc)[0]
    assert one == two
    sample = 1 << 13
    assert ctx.run(sc[:sample].contiguous(), npoints=sample)[0] == _oracle_g2(
        oracle, np.ascontiguousarray(np.tile(tile, (sample // distinct, 1))), sc[:sample].cpu().numpy(), sample, cid)
    ctx.close()


def test_option_is_ignored_on_g1(ea, oracle, torch_cuda):
    from conftest import oracle_msm_np

    n = 2000
    bases = ea.generate_points(n, distinct=64, seed=3, curve="bls12_381_g1")
    sc = rand_scalars_np(1, n, 8)
    ctx = ea.multi_scalar_mult_init(bases, "bls12_381_g1")
    ctx.set_option("g2_paired", ALL)
    assert ctx.run(sc)[0] == oracle_msm_np(oracle, 1, bases, sc, n)
    ctx.close()
