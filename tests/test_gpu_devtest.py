"""GPU: the DEVICE build of the arithmetic, element by element (reference counterpart
SPK ff/mont_t.cuh:385-425 mul/sqr, SPK ec/xyzz_t.hpp:178-249 mixed add).

fp28.hpp's multiply chains, Montgomery step, selects and quad permutes are inline GCN assembly / DPP on the device and
portable loops on the host, so tests/test_field_host.py never executes what the kernels execute, and whole MSMs on random data
never drive limbs to the lazy bounds (2^30 - 1 for fe_mul / fe_sqr, 2^29 - 1 for fe_mul2, the carried-operand classes of
Fp2El::mul_c) the 64-bit column argument depends on.  Here the SAME raw limb records go through

  * libmsm_devtest.so  (csrc/devtest.hip: one GPU thread -- or one quad -- per record), and
  * libmsm_hosttest.so (the same templates compiled by g++ with the limb-bound checker armed),

and the limbs that come back must be identical; the values are then checked against Python big integers (field ops) and the
affine chord-and-tangent model oracle/pymodel.py / oracle/te_model.py (group ops), which tests/test_oracle.py pins to the
reference.  2^16 records per field op, every record class below in each."""
import ctypes
import os
import random

import numpy as np
import pytest

import pymodel as m
from conftest import ROOT

pytestmark = pytest.mark.gpu

NL, LB = 14, 28
LMASK = (1 << LB) - 1
R392 = 1 << (NL * LB)
OPS = dict(FE_MUL=0, FE_SQR=1, FE_MUL2=2, NOT_AND_LMASK=3, EL_MUL=4, EL_SQR=5, EL_MUL_C=6, EL_MUL_C_BIG=7, EL_SQR_C=8,
           EL_MUL_SUB_C=9, MADD_COMMON=10, MADD=11, ADD=12, DBL=13, TE_MADD=14, TE_MADD_SWAPPED=15, TE_ADD=16, TE_DBL=17,
           ADD_QUAD=18, TE_ADD_QUAD=19, FE_WEAK_REDUCE=20)
DT_PAIR = 64
CURVES = {0: m.BLS12_377_G1, 1: m.BLS12_381_G1, 2: m.BLS12_377_G2, 3: m.BLS12_381_G2}


# ---- limb helpers ---------------------------------------------------------------------------------------------------
def limbs_of(v, top_free=True):
    """Normalized radix-2^28 limbs of a non-negative integer (the top limb takes what is left)."""
    out = [(v >> (LB * i)) & LMASK for i in range(NL - 1)]
    out.append(v >> (LB * (NL - 1)))
    assert out[-1] < (1 << 32)
    return out


def value_of(l):
    return sum(int(x) << (LB * i) for i, x in enumerate(l))


def arr(records):
    return np.ascontiguousarray(np.array(records, dtype=np.uint32))


def load_libs(with_device):
    dev = ctypes.CDLL(os.path.join(ROOT, "2022-entries_amd", "libmsm_devtest.so")) if with_device else None
    host = ctypes.CDLL(os.path.join(ROOT, "2022-entries_amd", "libmsm_hosttest.so"))
    host.ht_first_failure.restype = ctypes.c_char_p
    host.ht_check_failures.restype = ctypes.c_long
    if dev is not None:
        dev.msm_devtest_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        dev.msm_devtest_shape.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    host.ht_devop.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    host.ht_devop_shape.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    host.ht_reset_checks()
    return dev, host


@pytest.fixture(scope="module")
def libs(built):
    return load_libs(True)


def run_both(libs, cid, op, records, host_too=True):
    """records: uint32 [n, in_words] -> (device out, host out) as uint32 [n, out_words]; asserts they are identical and that the
    host's limb-bound checker saw nothing."""
    dev, host = libs
    iw, ow = ctypes.c_int(), ctypes.c_int()
    assert host.ht_devop_shape(cid, OPS[op], ctypes.byref(iw), ctypes.byref(ow)) == 0, (cid, op)
    assert records.dtype == np.uint32 and records.shape[1] == iw.value, (records.shape, iw.value)
    n = records.shape[0]
    out_d = None
    if dev is not None:
        di, do = ctypes.c_int(), ctypes.c_int()
        assert dev.msm_devtest_shape(cid, OPS[op], ctypes.byref(di), ctypes.byref(do)) == 0 and (di.value, do.value) == (iw.value, ow.value)
        out_d = np.zeros((n, ow.value), dtype=np.uint32)
        assert dev.msm_devtest_run(cid, OPS[op], records.ctypes.data, out_d.ctypes.data, n) == 0
    if not host_too:
        return out_d, None
    out_h = np.zeros((n, ow.value), dtype=np.uint32)
    host.ht_reset_checks()
    assert host.ht_devop(cid, OPS[op], records.ctypes.data, out_h.ctypes.data, n) == 0
    assert host.ht_check_failures() == 0, (op, host.ht_first_failure())
    if dev is None:            # tests/test_devtest_host.py: the host half alone (CPU suite)
        return out_h, out_h
    bad = np.nonzero((out_d != out_h).any(axis=1))[0]
    assert bad.size == 0, f"{op} curve {cid}: device and host limbs differ on {bad.size} of {n} records, first {bad[0]}: in={records[bad[0]].tolist()}"
    if cid >= 2 and OPS["EL_MUL"] <= OPS[op] <= OPS["DBL"]:
        # G2: the same records through the two-lanes-per-point form (csrc/fp2pair.hpp: lane h of a pair holds half h of every Fp2
        # value, partner limbs by DPP) -- the product kernels behind option "g2_paired" -- must give the same LIMBS
        out_p = np.full((n, ow.value), 0xEEEEEEEE, dtype=np.uint32)
        assert dev.msm_devtest_run(cid, DT_PAIR + OPS[op], records.ctypes.data, out_p.ctypes.data, n) == 0
        bad = np.nonzero((out_p != out_h).any(axis=1))[0]
        assert bad.size == 0, f"{op} curve {cid}, paired lanes: limbs differ from the host on {bad.size} of {n} records, first {bad[0]}: in={records[bad[0]].tolist()}"
    return out_d, out_h


# ---- field-operand classes ------------------------------------------------------------------------------------------------
def fe_classes(p, rng, n, limb_bits, val_mult):
    """n limb records for a multiplier whose contract is `limbs < 2^limb_bits, value < val_mult * p`: canonical values incl.
    0, 1, p - 1; p, 2p - 1 (class M's edge); every limb at 2^limb_bits - 1 with the top limb capped so the VALUE bound holds;
    single hot limbs; sparse patterns (a zero low limb makes m_k = 0 in the Montgomery step); lazy sums of canonical values
    written limb-wise (unnormalized); random limbs below the bound."""
    # largest top limb that keeps value < val_mult * p when the thirteen limbs below are maximal (they add up to
    # ~2^(limb_bits - 28) units of the top limb's weight)
    top_cap = ((val_mult * p) >> (LB * (NL - 1))) - (1 << (limb_bits - LB)) - 1
    lim = (1 << limb_bits) - 1
    recs = [limbs_of(v) for v in (0, 1, p - 1, p, 2 * p - 1, p + 1, (1 << LB) - 1, 1 << LB, R392 % p)]
    recs.append([lim] * (NL - 1) + [min(lim, top_cap - 1)])
    recs.append([lim] * (NL - 1) + [0])
    recs.append([0] * (NL - 1) + [min(lim, top_cap - 1)])
    for i in range(NL - 1):
        r = [0] * NL
        r[i] = lim
        recs.append(r)
        r = [lim] * (NL - 1) + [min(lim, top_cap - 1)]
        r[i] = 0
        recs.append(r)
    while len(recs) < n:
        k = rng.randrange(4)
        if k == 0:
            recs.append(limbs_of(rng.randrange(p)))
        elif k == 1:      # lazy limb-wise sum of up to (2^limb_bits / 2^28) canonical values
            terms = rng.randrange(1, max(2, min(val_mult, 1 << (limb_bits - LB))))
            acc = [0] * NL
            for _ in range(terms):
                acc = [a + b for a, b in zip(acc, limbs_of(rng.randrange(p)))]
            recs.append(acc)
        elif k == 2:      # random limbs right below the bound
            r = [rng.randrange(lim - 1000, lim + 1) for _ in range(NL - 1)] + [rng.randrange(0, min(lim, top_cap - 1) + 1)]
            recs.append(r)
        else:             # uniformly random limbs
            recs.append([rng.randrange(lim + 1) for _ in range(NL - 1)] + [rng.randrange(0, min(lim, top_cap - 1) + 1)])
    return recs[:n]


def check_class_m(out, p):
    """class M: limbs 0..12 < 2^28, value < 2p."""
    assert (out[:, : NL - 1] <= LMASK).all()
    for r in out[:: max(1, len(out) // 4096)]:
        assert value_of(r) < 2 * p


@pytest.mark.parametrize("cid", [0, 1])
def test_fe_mul_sqr_mul2_at_the_lazy_bounds(libs, cid):
    p = CURVES[cid].p
    rng = random.Random(100 + cid)
    n = 1 << 16
    rinv = pow(R392, -1, p)
    # fe_mul / fe_sqr: limbs < 2^30; value(a) * value(b) <= 2^10 p^2 -> both < 32p
    a, b = fe_classes(p, rng, n, 30, 32), fe_classes(p, rng, n, 30, 32)
    rng.shuffle(b)
    b[:64] = a[:64]                      # the extreme records against each other, too
    out, _ = run_both(libs, cid, "FE_MUL", arr([x + y for x, y in zip(a, b)]))
    check_class_m(out, p)
    for i in list(range(96)) + [rng.randrange(n) for _ in range(3000)]:
        assert value_of(out[i]) % p == value_of(a[i]) * value_of(b[i]) * rinv % p, i
    out, _ = run_both(libs, cid, "FE_SQR", arr(a))
    check_class_m(out, p)
    for i in list(range(96)) + [rng.randrange(n) for _ in range(3000)]:
        assert value_of(out[i]) % p == value_of(a[i]) ** 2 * rinv % p, i
    # fe_mul2: limbs < 2^29; a*b + c*d <= 2^10 p^2 -> all four < 22p
    q = [fe_classes(p, rng, n, 29, 22) for _ in range(4)]
    for v in q[1:]:
        rng.shuffle(v)
        v[:64] = q[0][:64]
    out, _ = run_both(libs, cid, "FE_MUL2", arr([w + x + y + z for w, x, y, z in zip(*q)]))
    check_class_m(out, p)
    for i in list(range(96)) + [rng.randrange(n) for _ in range(3000)]:
        assert value_of(out[i]) % p == (value_of(q[0][i]) * value_of(q[1][i]) + value_of(q[2][i]) * value_of(q[3][i])) * rinv % p, i


@pytest.mark.parametrize("cid", [0, 1])
def test_weak_reduce_and_bfi_step(libs, cid):
    p = CURVES[cid].p
    rng = random.Random(7 + cid)
    recs = fe_classes(p, rng, 1 << 14, 31, 32)
    out, _ = run_both(libs, cid, "FE_WEAK_REDUCE", arr(recs))
    assert (out[:, : NL - 1] <= LMASK).all()
    for i in range(0, len(recs), 7):
        v = value_of(out[i])
        assert v < 3 * p and v % p == value_of(recs[i]) % p
    words = [0, 1, LMASK, LMASK + 1, 0xFFFFFFFF, 0xF0000000, 0x0FFFFFFE] + [rng.randrange(1 << 32) for _ in range(4089)]
    out, _ = run_both(libs, cid, "NOT_AND_LMASK", arr([[w] for w in words]))
    assert [int(x) for x in out[:, 0]] == [(~w) & LMASK for w in words]


# ---- Fp2 (G2) carried-operand forms ----------------------------------------------------------------------------------
def carried(p, rng, n, val_mult):
    """Carried operands (what fe_carry leaves): limbs < 2^28 + 16, value < val_mult * p."""
    recs = []
    top_cap = ((val_mult * p) >> (LB * (NL - 1))) - 2
    lim = (1 << LB) + 15
    recs.append([lim] * (NL - 1) + [top_cap - 1])
    recs.append([0] * NL)
    recs.append(limbs_of(p - 1))
    recs.append(limbs_of(p))
    while len(recs) < n:
        k = rng.randrange(3)
        if k == 0:
            recs.append(limbs_of(rng.randrange(min(val_mult, 2) * p)))
        elif k == 1:
            recs.append([rng.randrange(lim - 40, lim + 1) for _ in range(NL - 1)] + [rng.randrange(top_cap)])
        else:
            v = limbs_of(rng.randrange((val_mult - 1) * p))
            recs.append([x + rng.randrange(16) for x in v[:-1]] + [v[-1]])
    return recs[:n]


def class_m(p, rng, n):
    """class M as fp28.hpp defines it: a multiplier's output, strictly normalized, value < p + a b / R < 1.5 p (the biased
    subtractions that take a class-M operand -- BIAS2_28 -- rely on the top limb that bound implies)."""
    recs = [limbs_of(v) for v in (0, 1, p - 1, p, 3 * p // 2 - 1)]
    while len(recs) < n:
        recs.append(limbs_of(rng.randrange(3 * p // 2)))
    return recs[:n]


def fp2_val(c, rec):
    return c.F((value_of(rec[:NL]), value_of(rec[NL:])))


@pytest.mark.parametrize("cid", [2, 3])
def test_fp2_products_in_every_operand_class(libs, cid):
    c = CURVES[cid]
    p = c.p
    rng = random.Random(20 + cid)
    n = 1 << 14
    rinv = pow(R392, -1, p)

    def pairs(mk_a, mk_b):
        a0, a1, b0, b1 = mk_a(), mk_a(), mk_b(), mk_b()
        for v in (a1, b1):
            rng.shuffle(v)
        a1[:4] = a0[:4]          # the extreme records of a class in both components at once
        b1[:4] = b0[:4]
        return [w + x for w, x in zip(a0, a1)], [w + x for w, x in zip(b0, b1)]

    def check(op, a, b, out):
        for i in list(range(8)) + [rng.randrange(n) for _ in range(1500)]:
            exp = fp2_val(c, a[i]) * (fp2_val(c, b[i]) if b is not None else fp2_val(c, a[i])) * rinv
            got = fp2_val(c, out[i])
            assert got == exp, (op, i)
        check_class_m(out[:, :NL], p)
        check_class_m(out[:, NL:], p)

    # E::mul: any lazy operands (limbs < 2^30 after the formulas' additions; values <= 18p)
    a, b = pairs(lambda: fe_classes(p, rng, n, 30, 18), lambda: fe_classes(p, rng, n, 30, 18))
    out, _ = run_both(libs, cid, "EL_MUL", arr([x + y for x, y in zip(a, b)]))
    check("EL_MUL", a, b, out)
    out, _ = run_both(libs, cid, "EL_SQR", arr(a))
    check("EL_SQR", a, None, out)
    # mul_c<false>: a carried (<= 18p), b class M;  mul_c<true>: both carried, b <= 18p;  sqr_c: a carried
    a, b = pairs(lambda: carried(p, rng, n, 18), lambda: class_m(p, rng, n))
    out, _ = run_both(libs, cid, "EL_MUL_C", arr([x + y for x, y in zip(a, b)]))
    check("EL_MUL_C", a, b, out)
    a, b = pairs(lambda: carried(p, rng, n, 18), lambda: carried(p, rng, n, 18))
    out, _ = run_both(libs, cid, "EL_MUL_C_BIG", arr([x + y for x, y in zip(a, b)]))
    check("EL_MUL_C_BIG", a, b, out)
    out, _ = run_both(libs, cid, "EL_SQR_C", arr(a))
    check("EL_SQR_C", a, None, out)
    # mul_sub_c: a*b - c*d;  a, b, c carried (b <= 18p), d class M
    a, b = pairs(lambda: carried(p, rng, n, 18), lambda: carried(p, rng, n, 18))
    cc, d = pairs(lambda: carried(p, rng, n, 18), lambda: class_m(p, rng, n))
    out, _ = run_both(libs, cid, "EL_MUL_SUB_C", arr([w + x + y + z for w, x, y, z in zip(a, b, cc, d)]))
    for i in list(range(8)) + [rng.randrange(n) for _ in range(1000)]:
        exp = (fp2_val(c, a[i]) * fp2_val(c, b[i]) - fp2_val(c, cc[i]) * fp2_val(c, d[i])) * rinv
        assert fp2_val(c, out[i]) == exp, i
    assert (out <= LMASK + 16).all()


# ---- group law ---------------------------------------------------------------------------------------------------------
def mont(c, v):
    """coordinate-field value -> internal Montgomery residue(s) (R = 2^392), canonical, as limb list(s)."""
    if c.ext == 1:
        return limbs_of(v * R392 % c.p)
    return limbs_of(v.c0 * R392 % c.p) + limbs_of(v.c1 * R392 % c.p)


def lift(c, limbs, k, rng):
    """the same residue as a stored X / Y coordinate may hold it: + k p, re-normalized (value < 16p, limbs < 2^28 + 16)"""
    outl = []
    for j in range(c.ext):
        v = value_of(limbs[j * NL:(j + 1) * NL]) + k * c.p
        l = limbs_of(v)
        outl += l
    return outl


def unmont(c, limbs):
    rinv = pow(R392, -1, c.p)
    if c.ext == 1:
        return value_of(limbs) * rinv % c.p
    return c.F((value_of(limbs[:NL]) * rinv, value_of(limbs[NL:]) * rinv))


def xyzz_record(c, P, rng, lift_k=0):
    """affine P (None = infinity) as an XYZZ record with a random Z: X = x z^2, Y = y z^3, ZZ = z^2, ZZZ = z^3."""
    ew = NL * c.ext
    if P is None:
        z = c.F(0 if c.ext == 1 else (0, 0))
        x, y = c.F(rng.randrange(c.p) if c.ext == 1 else (rng.randrange(c.p), 0)), c.F(1 if c.ext == 1 else (1, 0))
        return mont(c, x) + mont(c, y) + [0] * (2 * ew)
    z = c.F(rng.randrange(1, c.p) if c.ext == 1 else (rng.randrange(1, c.p), rng.randrange(c.p)))
    zz = c.f_mul(z, z)
    zzz = c.f_mul(zz, z)
    X, Y = c.f_mul(P[0], zz), c.f_mul(P[1], zzz)
    return lift(c, mont(c, X), lift_k, rng) + lift(c, mont(c, Y), lift_k, rng) + mont(c, zz) + mont(c, zzz)


def xyzz_to_affine(c, rec):
    ew = NL * c.ext
    X, Y, ZZ, ZZZ = (unmont(c, rec[i * ew:(i + 1) * ew]) for i in range(4))
    if c.f_is_zero(ZZ):
        return None
    return (c.f_mul(X, c.f_inv(ZZ)), c.f_mul(Y, c.f_inv(ZZZ)))


def point_cases(c, rng, n):
    """(P, Q) pairs: general, P == Q (doubling), P == -Q (cancellation), infinity on either side or both."""
    pts = m.random_points(c, 48, rng)
    cases = [(pts[0], pts[0]), (pts[1], c.neg(pts[1])), (None, pts[2]), (pts[3], None), (None, None)]
    while len(cases) < n:
        cases.append((rng.choice(pts), rng.choice(pts)))
    return cases


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_xyzz_additions_match_the_affine_model(libs, cid):
    c = CURVES[cid]
    rng = random.Random(300 + cid)
    n = 384 if cid < 2 else 160
    ew = NL * c.ext
    cases = point_cases(c, rng, n)
    # full addition, one lane and four lanes per record; stored X / Y lifted by up to 14 p (the invariant allows < 16p)
    recs = arr([xyzz_record(c, P, rng, rng.choice((0, 0, 1, 7, 14))) + xyzz_record(c, Q, rng, rng.choice((0, 0, 1, 7, 14))) for P, Q in cases])
    out, _ = run_both(libs, cid, "ADD", recs)
    quad, _ = run_both(libs, cid, "ADD_QUAD", recs, host_too=False)
    for i, (P, Q) in enumerate(cases):
        exp = c.add(P, Q)
        assert xyzz_to_affine(c, out[i].tolist()) == exp, ("ADD", i)
        assert quad is None or xyzz_to_affine(c, quad[i].tolist()) == exp, ("ADD_QUAD", i)
    # doubling
    recs = arr([xyzz_record(c, P, rng, rng.choice((0, 3, 14))) for P, _ in cases if P is not None])
    out, _ = run_both(libs, cid, "DBL", recs)
    for i, P in enumerate([P for P, _ in cases if P is not None]):
        assert xyzz_to_affine(c, out[i].tolist()) == c.add(P, P), ("DBL", i)
    # mixed addition: every case through MADD; the common path alone must report "same x" exactly for P == +/-Q
    mixed = [(P, Q) for P, Q in cases if Q is not None]
    recs, flags = [], []
    for P, Q in mixed:
        neg = rng.randrange(2)
        fresh = 1 if P is None and rng.randrange(2) else 0          # acc_inf: the caller knows the accumulator is empty
        base = mont(c, Q[0]) + mont(c, Q[1])
        recs.append(xyzz_record(c, P, rng, rng.choice((0, 2, 14))) + base + [neg | (fresh << 1)])
        flags.append((neg, fresh))
    recs = arr(recs)
    out, _ = run_both(libs, cid, "MADD", recs)
    com, _ = run_both(libs, cid, "MADD_COMMON", recs)
    for i, (P, Q) in enumerate(mixed):
        Qs = c.neg(Q) if flags[i][0] else Q
        exp = c.add(P, Qs)
        assert xyzz_to_affine(c, out[i, :4 * ew].tolist()) == exp, ("MADD", i)
        same_x = P is not None and P[0] == Q[0]
        assert int(com[i, 4 * ew]) == (1 if same_x else 0), ("MADD_COMMON flag", i)
        if not same_x:
            assert xyzz_to_affine(c, com[i, :4 * ew].tolist()) == exp, ("MADD_COMMON", i)
        else:
            assert (com[i, :4 * ew] == recs[i, :4 * ew]).all()       # untouched: the caller re-reads the base and finishes


def test_twisted_edwards_additions_match_the_model(libs):
    import te_model as te

    c = CURVES[0]
    p = c.p
    rng = random.Random(55)
    pts = m.random_points(c, 40, rng)
    timg = [te.sw_to_te(P) for P in pts]
    assert all(t is not None for t in timg)

    def ext_record(t):          # affine TE (X, Y) -> extended (X z, Y z, z, X Y z), class M residues
        z = rng.randrange(1, p)
        X, Y = t
        vals = (X * z % p, Y * z % p, z, X * Y % p * z % p)
        out = []
        for v in vals:
            out += limbs_of(v * R392 % p + (p if rng.randrange(4) == 0 and v * R392 % p + p < 3 * p // 2 else 0))
        return out

    def ext_to_affine(rec):
        X, Y, Z, T = (unmont(c, rec[i * NL:(i + 1) * NL]) for i in range(4))
        assert Z != 0 and X * Y % p == Z * T % p
        zi = pow(Z, -1, p)
        return (X * zi % p, Y * zi % p)

    ident = (0, 1)
    cases = [(timg[0], timg[0]), (timg[1], te.te_neg(timg[1])), (ident, timg[2]), (timg[3], ident), (ident, ident)]
    while len(cases) < 320:
        cases.append((rng.choice(timg), rng.choice(timg)))
    recs = arr([ext_record(a) + ext_record(b) for a, b in cases])
    out, _ = run_both(libs, 0, "TE_ADD", recs)
    quad, _ = run_both(libs, 0, "TE_ADD_QUAD", recs, host_too=False)
    for i, (a, b) in enumerate(cases):
        exp = te.te_add(a, b)
        assert ext_to_affine(out[i].tolist()) == exp, ("TE_ADD", i)
        assert quad is None or ext_to_affine(quad[i].tolist()) == exp, ("TE_ADD_QUAD", i)
    check_class_m(out.reshape(-1, NL), p)
    if quad is not None:
        check_class_m(quad.reshape(-1, NL), p)
    out, _ = run_both(libs, 0, "TE_DBL", arr([ext_record(a) for a, _ in cases]))
    for i, (a, _) in enumerate(cases):
        assert ext_to_affine(out[i].tolist()) == te.te_add(a, a), ("TE_DBL", i)
    # mixed addition against the device base record (Y - X, Y + X, 2 d X Y), both forms of the negation
    recs, recs_sw, negs = [], [], []
    for a, b in cases:
        neg = rng.randrange(2)
        ymx, ypx, td = te.te_precomp(b)
        base = limbs_of(ymx * R392 % p) + limbs_of(ypx * R392 % p) + limbs_of(td * R392 % p)
        base_sw = (limbs_of(ypx * R392 % p) + limbs_of(ymx * R392 % p) if neg else base[:2 * NL]) + base[2 * NL:]
        acc = ext_record(a)
        recs.append(acc + base + [neg])
        recs_sw.append(acc + base_sw + [neg])
        negs.append(neg)
    out, _ = run_both(libs, 0, "TE_MADD", arr(recs))
    out_sw, _ = run_both(libs, 0, "TE_MADD_SWAPPED", arr(recs_sw))
    assert (out == out_sw).all()          # the k_accumulate_glds form (operands pre-swapped by LDS address) is the same addition
    for i, (a, b) in enumerate(cases):
        exp = te.te_add(a, te.te_neg(b) if negs[i] else b)
        assert ext_to_affine(out[i].tolist()) == exp, ("TE_MADD", i)
