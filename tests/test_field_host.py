"""CPU: the kernels' own arithmetic (fp28.hpp / curve.hpp, radix-2^28 lazy reduction) compiled for the host with the
limb-bound checker armed, against the Python model.  This is where the lazy-reduction bounds are proven not to overflow."""
import ctypes
import os
import random

import pytest

import pymodel as m
from conftest import ROOT

CURVES = [(0, m.BLS12_377_G1), (1, m.BLS12_381_G1)]


@pytest.fixture(scope="module")
def ht(built):
    lib = ctypes.CDLL(os.path.join(ROOT, "2022-entries_amd", "libmsm_hosttest.so"))
    lib.ht_first_failure.restype = ctypes.c_char_p
    lib.ht_check_failures.restype = ctypes.c_long
    for name in ("ht_madd_chain", "ht_add_chains", "ht_msm_naive"):
        getattr(lib, name).restype = ctypes.c_int
    lib.ht_madd_chain.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    lib.ht_add_chains.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p]
    lib.ht_msm_naive.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    lib.ht_reset_checks()
    yield lib
    assert lib.ht_check_failures() == 0, lib.ht_first_failure()


@pytest.mark.parametrize("cid,curve", CURVES)
def test_field_mul_roundtrip_inverse(ht, cid, curve):
    p = curve.p
    rng = random.Random(42 + cid)
    rinv = pow(m.R, -1, p)
    samples = [(0, 5), (p - 1, p - 1), (1, p - 1), (m.R % p, m.R % p)] + [(rng.randrange(p), rng.randrange(p)) for _ in range(500)]
    out = ctypes.create_string_buffer(48)
    for a, b in samples:
        ht.ht_fe_mul(cid, a.to_bytes(48, "little"), b.to_bytes(48, "little"), out)
        assert int.from_bytes(out.raw, "little") == (a * b * rinv) % p
        ht.ht_fe_roundtrip(cid, a.to_bytes(48, "little"), out)
        assert int.from_bytes(out.raw, "little") == a
        ht.ht_fe_sqr(cid, a.to_bytes(48, "little"), out)
        assert int.from_bytes(out.raw, "little") == (a * a * rinv) % p
    # every limb at the largest value a multiply may see: only the column-overflow checker matters here
    assert ht.ht_fe_extreme(cid) == 0
    for _ in range(5):
        a = rng.randrange(1, p)
        ht.ht_fe_inv(cid, a.to_bytes(48, "little"), out)
        x = (a * rinv) % p
        assert int.from_bytes(out.raw, "little") == (pow(x, -1, p) * m.R) % p
    assert ht.ht_check_failures() == 0, ht.ht_first_failure()


@pytest.mark.parametrize("cid,curve", CURVES)
def test_mixed_add_all_branches(ht, cid, curve):
    """first element, general add, doubling (P + P), cancellation (P - P), negated inputs, infinity inputs."""
    rng = random.Random(7 + cid)
    pts = m.random_points(curve, 12, rng)
    out = ctypes.create_string_buffer(144)
    seq = [pts[0], pts[0], pts[1], curve.neg(pts[1]), pts[2], None, pts[3], pts[3], pts[3]]
    negs = [0, 0, 1, 1, 0, 0, 1, 1, 0]
    exp = None
    for P, ng in zip(seq, negs):
        exp = curve.add(exp, curve.neg(P) if ng else P)
    ht.ht_madd_chain(cid, curve.encode_affine_array(seq), 104, bytes(negs), len(seq), out)
    assert out.raw == curve.encode_projective_normalized(exp)
    ht.ht_madd_chain(cid, curve.encode_affine_array([pts[5], pts[5]]), 104, bytes([0, 1]), 2, out)
    assert out.raw == curve.encode_projective_normalized(None)
    # long random chains keep the lazy bounds honest
    for _ in range(20):
        k = rng.randrange(2, 40)
        seq = [pts[rng.randrange(12)] for _ in range(k)]
        negs = [rng.randrange(2) for _ in range(k)]
        exp = None
        for P, ng in zip(seq, negs):
            exp = curve.add(exp, curve.neg(P) if ng else P)
        ht.ht_madd_chain(cid, curve.encode_affine_array(seq), 104, bytes(negs), k, out)
        assert out.raw == curve.encode_projective_normalized(exp)
    assert ht.ht_check_failures() == 0, ht.ht_first_failure()


@pytest.mark.parametrize("cid,curve", CURVES)
def test_full_add_and_naive_msm(ht, cid, curve):
    rng = random.Random(9 + cid)
    pts = m.random_points(curve, 8, rng)
    out = ctypes.create_string_buffer(144)
    seq = [pts[0], pts[1], pts[0], pts[1]]  # equal operands -> doubling inside the full add
    ht.ht_add_chains(cid, curve.encode_affine_array(seq), 104, 2, 2, out)
    assert out.raw == curve.encode_projective_normalized(curve.mul(2, curve.add(pts[0], pts[1])))
    seq = [pts[0], pts[1], curve.neg(curve.add(pts[0], pts[1]))]  # opposite operands -> infinity
    ht.ht_add_chains(cid, curve.encode_affine_array(seq), 104, 2, 1, out)
    assert out.raw == curve.encode_projective_normalized(None)
    seq = [pts[0], pts[1], pts[2]]
    for na in (0, 1, 2, 3):
        ht.ht_add_chains(cid, curve.encode_affine_array(seq), 104, na, 3 - na, out)
        assert out.raw == curve.encode_projective_normalized(curve.add(curve.add(pts[0], pts[1]), pts[2]))
    sc = m.random_scalars(curve, 8, rng)
    sc[2], sc[3], sc[4] = 0, 1, (1 << 256) - 1
    ht.ht_msm_naive(cid, curve.encode_affine_array(pts), 104, m.encode_scalars(sc), 8, out)
    assert out.raw == curve.encode_projective_normalized(curve.msm_naive(pts, sc))
    assert ht.ht_check_failures() == 0, ht.ht_first_failure()


def test_two_torsion_and_edge_points(ht):
    """T = (p-1, 0) has order 2: T + T must come out as infinity through the doubling formula, and a later add onto
    that accumulator must treat it as infinity (P1B msm_unit_tests.rs:21-77)."""
    c = m.BLS12_377_G1
    out = ctypes.create_string_buffer(144)
    for seq in ([m.EDGE_P, m.EDGE_P_NEG, m.EDGE_T, m.EDGE_T], [m.EDGE_T, m.EDGE_T, m.EDGE_P], [m.EDGE_T, m.EDGE_P],
                [m.EDGE_T, m.EDGE_P, m.EDGE_T]):
        exp = None
        for P in seq:
            exp = c.add(exp, P)
        ht.ht_madd_chain(0, c.encode_affine_array(seq), 104, bytes(len(seq)), len(seq), out)
        assert out.raw == c.encode_projective_normalized(exp)
    assert ht.ht_check_failures() == 0, ht.ht_first_failure()


# ---- G2: Fp2 coordinates (BLS12-377, u^2 = -5) through the same generic curve templates ----------------------------

def _fp2_enc(c, v):
    return ((v.c0 * m.R) % c.p).to_bytes(48, "little") + ((v.c1 * m.R) % c.p).to_bytes(48, "little")


def _fp2_dec(c, b):
    ri = pow(m.R, -1, c.p)
    return m.Fp2((int.from_bytes(b[:48], "little") * ri) % c.p, (int.from_bytes(b[48:96], "little") * ri) % c.p, c.p, c.nonresidue % c.p)


def test_weak_reduce_and_fp2_field(ht):
    rng = random.Random(8)
    for cid, c in CURVES:
        out = ctypes.create_string_buffer(48)
        for t in range(300):
            a, k = rng.randrange(c.p), rng.choice([1, 2, 3, 5, 7, 14, 21, 28])
            if t < 3:
                a, k = c.p - 1, 28
            assert ht.ht_fe_weak_reduce(cid, a.to_bytes(48, "little"), k, out) == 0   # also asserts result < 3p
            assert int.from_bytes(out.raw, "little") == (k * a) % c.p
    for cid, c in ((2, m.BLS12_377_G2), (3, m.BLS12_381_G2)):
        out = ctypes.create_string_buffer(96)
        for _ in range(200):
            a = m.Fp2(rng.randrange(c.p), rng.randrange(c.p), c.p, c.nonresidue % c.p)
            b = m.Fp2(rng.randrange(c.p), rng.randrange(c.p), c.p, c.nonresidue % c.p)
            ht.ht_el_mul(cid, _fp2_enc(c, a), _fp2_enc(c, b), out)
            assert _fp2_dec(c, out.raw) == a * b
            ht.ht_el_inv(cid, _fp2_enc(c, a), out)
            assert _fp2_dec(c, out.raw) == a.inv()
    assert ht.ht_check_failures() == 0, ht.ht_first_failure()


@pytest.mark.parametrize("cid,c", [(2, m.BLS12_377_G2), (3, m.BLS12_381_G2)], ids=["bls12_377_g2", "bls12_381_g2"])
def test_g2_group_law(ht, cid, c):
    rng = random.Random(12)
    pts = m.random_points(c, 10, rng)
    out = ctypes.create_string_buffer(288)
    seq = [pts[0], pts[0], pts[1], c.neg(pts[1]), pts[2], None, pts[3], pts[3], pts[3]]
    negs = [0, 0, 1, 1, 0, 0, 1, 1, 0]
    exp = None
    for P, ng in zip(seq, negs):
        exp = c.add(exp, c.neg(P) if ng else P)
    ht.ht_madd_chain(cid, c.encode_affine_array(seq), 200, bytes(negs), len(seq), out)
    assert out.raw == c.encode_projective_normalized(exp)
    ht.ht_add_chains(cid, c.encode_affine_array([pts[0], pts[1], pts[0], pts[1]]), 200, 2, 2, out)
    assert out.raw == c.encode_projective_normalized(c.mul(2, c.add(pts[0], pts[1])))
    ht.ht_add_chains(cid, c.encode_affine_array([pts[0], c.neg(pts[0])]), 200, 1, 1, out)
    assert out.raw == c.encode_projective_normalized(None)
    sc = m.random_scalars(c, 6, rng)
    sc[1], sc[2] = 0, 1
    ht.ht_msm_naive(cid, c.encode_affine_array(pts[:6]), 200, m.encode_scalars(sc), 6, out)
    assert out.raw == c.encode_projective_normalized(c.msm_naive(pts[:6], sc))
    assert ht.ht_check_failures() == 0, ht.ht_first_failure()
