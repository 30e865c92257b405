"""CPU: the 64-bit host tail of an MSM (csrc/host_fold64.hpp: window fold, chunk sums, normalisation on 6 x u64 Montgomery
limbs) against (i) the generic radix-2^28 arithmetic it replaces and (ii) the big-int model -- field operations, the
short-Weierstrass fold for both G1 curves, and the twisted-Edwards fold with the map back (BLS12-377)."""
import ctypes
import os
import random

import pytest

import pymodel as m
from conftest import ROOT

R384 = 1 << 384


@pytest.fixture(scope="module")
def ht(built):
    lib = ctypes.CDLL(os.path.join(ROOT, "2022-entries_amd", "libmsm_hosttest.so"))
    lib.ht_f64_op.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    lib.ht_fold_both.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
    return lib


def mont(v, p):
    return ((v * R384) % p).to_bytes(48, "little")


@pytest.mark.parametrize("cid,curve", [(0, m.BLS12_377_G1), (1, m.BLS12_381_G1)])
def test_f64_field_ops(ht, cid, curve):
    p = curve.p
    rng = random.Random(5 + cid)
    vals = [0, 1, p - 1, p - 2, 2, (p + 1) // 2] + [rng.randrange(p) for _ in range(40)]
    out = ctypes.create_string_buffer(48)
    for a in vals:
        for b in (vals[0], vals[2], rng.choice(vals), rng.randrange(p)):
            # op 0 = Fp64::mul (mulx / adcx / adox assembly where the CPU has them), op 4 = the portable no-carry CIOS loop
            for op, fn in ((0, lambda x, y: x * y), (4, lambda x, y: x * y), (1, lambda x, y: x + y), (2, lambda x, y: x - y)):
                assert ht.ht_f64_op(cid, op, mont(a, p), mont(b, p), out) == 0
                assert out.raw == mont(fn(a, b) % p, p), (op, a, b)
        if a:
            assert ht.ht_f64_op(cid, 3, mont(a, p), mont(0, p), out) == 0
            assert out.raw == mont(pow(a, p - 2, p), p)


@pytest.mark.parametrize("cid,curve,te", [(0, m.BLS12_377_G1, 0), (1, m.BLS12_381_G1, 0), (0, m.BLS12_377_G1, 1), (2, m.BLS12_377_G2, 0), (3, m.BLS12_381_G2, 0)])
def test_fold64_matches_generic_and_model(ht, cid, curve, te):
    rng = random.Random(31 + 2 * cid + te)
    pb = 288 if cid >= 2 else 144                      # G2 (Fp2_64 over Fp64): coordinates are c0 | c1
    for windows, c in ((1, 5), (3, 7), (13, 20), (37, 7), (11, 24)) if cid < 2 else ((1, 5), (3, 7), (13, 9)):
        pts = m.random_points(curve, windows, rng)
        mult = [rng.randrange(1, 1 << 40) for _ in range(windows)]
        if windows >= 3:
            mult[1] = 0                      # an empty window
            if not te:
                pts[2] = None                # a window sum at infinity
            pts[-1] = pts[0]                 # equal points: the add-becomes-doubling branch
            mult[-1] = mult[0]
        arr = (ctypes.c_uint64 * windows)(*mult)
        og, oh = ctypes.create_string_buffer(pb), ctypes.create_string_buffer(pb)
        rc = ht.ht_fold_both(cid, curve.encode_affine_array(pts), 200 if cid >= 2 else 104, arr, windows, c, te, og, oh)
        assert rc == 0, rc
        acc = None
        for w in range(windows - 1, -1, -1):
            for _ in range(c):
                acc = curve.add(acc, acc)
            acc = curve.add(acc, curve.mul(mult[w], pts[w]) if pts[w] is not None else None)
        assert og.raw == oh.raw == curve.encode_projective_normalized(acc), (windows, c)


def test_fold64_cancellation_to_infinity(ht):
    curve = m.BLS12_377_G1
    P = m.random_points(curve, 1, random.Random(9))[0]
    pts = [P, curve.neg(P)]
    # 2^4 * (1 * -P) + 16 * P = O
    arr = (ctypes.c_uint64 * 2)(16, 1)
    for te in (0, 1):
        og, oh = ctypes.create_string_buffer(144), ctypes.create_string_buffer(144)
        assert ht.ht_fold_both(0, curve.encode_affine_array(pts), 104, arr, 2, 4, te, og, oh) == 0
        assert og.raw == oh.raw == curve.encode_projective_normalized(None)
