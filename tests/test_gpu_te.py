"""GPU: the twisted-Edwards fast path of BLS12-377 G1 (csrc/te.hpp) is an implementation detail that must never change a
result: same bytes as the XYZZ path and as the oracle; base sets with a point the map is undefined on stay on XYZZ; an
addition with a vanishing denominator (the curve's d is a square, so they exist off the prime-order subgroup) is detected
on the device and the run is repeated on XYZZ -- also from a context whose short-Weierstrass tables were dropped."""
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


def _bases_np(pts):
    return np.frombuffer(C.encode_affine_array(pts), dtype=np.uint8).reshape(len(pts), 104).copy()


def test_path_selection_and_identical_results(ea, oracle):
    n = 5000
    bases = ea.generate_points(n, distinct=700, seed=3)
    sc = _scalars(2 * n, 11)
    exp = [oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(sc[b * n:(b + 1) * n]), n) for b in range(2)]
    ctx = ea.multi_scalar_mult_init(bases, "bls12_377_g1")
    assert ctx.query("twisted_edwards") == 1 and ctx.query("bases") == n
    assert ctx.run(sc) == exp and ctx.query("twisted_edwards_fallbacks") == 0
    off = ea.MultiScalarMultContext("bls12_377_g1")
    off.set_option("twisted_edwards", 0)
    off.set_bases(bases)
    assert off.query("twisted_edwards") == 0 and off.run(sc) == exp
    # the option only exists for the curve that has the image
    c381 = ea.multi_scalar_mult_init(ea.generate_points(64, distinct=8, seed=1, curve="bls12_381_g1"), "bls12_381_g1")
    assert c381.query("twisted_edwards") == 0
    # one base without an image (the FPGA harness's 2-torsion point, or a u = -1 point) keeps the whole set on XYZZ
    for special in te.exceptional_points():
        b2 = bases.copy()
        b2[17] = np.frombuffer(C.encode_affine(special), dtype=np.uint8)
        ctx.set_bases(b2)
        assert ctx.query("twisted_edwards") == 0
        assert ctx.run(np.ascontiguousarray(sc[:n]))[0] == oracle_msm_np(oracle, 0, b2, np.ascontiguousarray(sc[:n]), n)
    # ... and a base flagged infinite does not (it is never gathered)
    b3 = bases.copy()
    b3[5, 96] = 1
    ctx.set_bases(b3)
    assert ctx.query("twisted_edwards") == 1
    assert ctx.run(np.ascontiguousarray(sc[:n]))[0] == oracle_msm_np(oracle, 0, b3, np.ascontiguousarray(sc[:n]), n)
    for c in (ctx, off, c381):
        c.close()


def _failing_pair():
    """Two mappable points whose Edwards sum has a vanishing denominator: R = P + E with E of even order."""
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


@pytest.mark.parametrize("precompute", [0, 1])
def test_vanishing_denominator_falls_back_to_xyzz(ea, oracle, precompute):
    R, Q = _failing_pair()
    rng = random.Random(4)
    n = 3000
    pts = m.random_points(C, 50, rng)
    seq = [pts[i % 50] for i in range(n)]
    seq[100], seq[2000] = R, Q
    bases = _bases_np(seq)
    sc = np.zeros((n, 32), dtype=np.uint8)
    sc[100] = sc[2000] = _scalars(1, 5)[0]   # equal scalars, everything else zero: R and Q are alone in their buckets
    ctx = ea.MultiScalarMultContext("bls12_377_g1")
    ctx.set_option("precompute", precompute)
    ctx.set_bases(bases)
    assert ctx.query("twisted_edwards") == 1   # both points have an image; only their SUM is the problem
    got = ctx.run(sc)[0]
    assert got == oracle_msm_np(oracle, 0, bases, sc, n)
    assert ctx.query("twisted_edwards_fallbacks") == 1
    # the context keeps working, and inputs that avoid the pair stay on the fast path
    sc2 = _scalars(n, 6)
    sc2[100] = 0
    assert ctx.run(sc2)[0] == oracle_msm_np(oracle, 0, bases, sc2, n)
    assert ctx.query("twisted_edwards_fallbacks") == 1
    ctx.close()


def test_even_order_inputs_without_failures(ea, oracle):
    """Points outside the prime-order subgroup are legal inputs; most additions among them are still defined."""
    rng = random.Random(12)
    n = 2048
    T = (C.p - 1, 0)
    pts = [C.add(P, T) for P in m.random_points(C, 40, rng)]      # each has order 2r
    assert all(te.sw_to_te(P) is not None for P in pts)
    bases = _bases_np([pts[i % 40] for i in range(n)])
    sc = _scalars(n, 8)
    ctx = ea.multi_scalar_mult_init(bases, "bls12_377_g1")
    assert ctx.query("twisted_edwards") == 1
    assert ctx.run(sc)[0] == oracle_msm_np(oracle, 0, bases, sc, n)
    ctx.close()
