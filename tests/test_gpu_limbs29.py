"""GPU: the 13 x 29 limb shape of BLS12-377 Fq (csrc/fp28.hpp, Bls12_377_Fq29) and the twisted-Edwards law on it (csrc/te.hpp) --
the arithmetic of the BLS12-377 G1 hot path since round 6 -- element by element on the DEVICE build against the host build
(limb-bound checker armed) and against Python big integers / oracle/te_model.py, at the operand classes the law produces.

Reference semantics kept: SPK ff/mont_t.cuh:385-425 (Montgomery product), the lazy-bound style of ML ff_dispatch_st.cuh:453-476.
"Curve id 4" of the two test libraries is this shape (csrc/devtest_ops.hpp DT_CURVE_TE29): records keep 14 words per field element,
word 13 is 0.  The worst-case column sums are PROVEN by tools/limb_bounds29.py (tests/test_limbs29_host.py runs it); here the
records sit at those bounds and the device must return exactly the host's limbs."""
import random

import numpy as np
import pytest

import pymodel as m
import test_gpu_devtest as g
from test_gpu_devtest import arr, run_both

pytestmark = pytest.mark.gpu
libs = g.libs   # the module-scoped fixture of test_gpu_devtest (device + host library)

CID = 4
N29, B29, NRED29, NW = 13, 29, 14, 14
MASK29 = (1 << B29) - 1
R406 = 1 << (B29 * NRED29)
P = m.BLS12_377_G1.p
P_TOP = P >> (B29 * (N29 - 1))


def limbs29(v):
    """normalized radix-2^29 limbs (the 13th takes what is left) + the zero pad word"""
    out = [(v >> (B29 * i)) & MASK29 for i in range(N29 - 1)] + [v >> (B29 * (N29 - 1))]
    assert out[-1] < (1 << 32)
    return out + [0] * (NW - N29)


def value29(l):
    assert all(int(x) == 0 for x in l[N29:])
    return sum(int(x) << (B29 * i) for i, x in enumerate(l[:N29]))


def operand_class(rng, n, limb_mult, top_cap):
    """n records with limbs 0..11 <= limb_mult * 2^29 - 1 and the top limb <= top_cap: the extremes, single hot / cold limbs, sparse
    patterns (a zero low limb makes m_k = 0), lazy sums of canonical values, random limbs at and below the bound."""
    lim = limb_mult * (1 << B29) - 1
    recs = [limbs29(v) for v in (0, 1, P - 1, P, P + 1, MASK29, 1 << B29, R406 % P)]
    recs.append([lim] * (N29 - 1) + [top_cap, 0])
    recs.append([lim] * (N29 - 1) + [0, 0])
    recs.append([0] * (N29 - 1) + [top_cap, 0])
    for i in range(N29 - 1):
        r = [0] * NW
        r[i] = lim
        recs.append(r)
        r = [lim] * (N29 - 1) + [top_cap, 0]
        r[i] = 0
        recs.append(r)
    while len(recs) < n:
        k = rng.randrange(3)
        if k == 0:
            recs.append(limbs29(rng.randrange(P)))
        elif k == 1:
            recs.append([rng.randrange(max(0, lim - 1000), lim + 1) for _ in range(N29 - 1)] + [rng.randrange(top_cap + 1), 0])
        else:
            recs.append([rng.randrange(lim + 1) for _ in range(N29 - 1)] + [rng.randrange(top_cap + 1), 0])
    return recs[:n]


def check_class_m29(out):
    assert (out[:, : N29 - 1] <= MASK29).all() and (out[:, N29:] == 0).all()
    for r in out[:: max(1, len(out) // 4096)]:
        assert value29(r) < P + (P >> 16)


# the operand classes of te_madd / te_tail / te_add, as tools/limb_bounds29.py derives them: (limbs a, top a, limbs b, top b)
CLASSES = [
    ("A = (Y1 - X1)(Y2 - X2)", 3, int(2.95 * 2**29), 1, P_TOP),
    ("B = (Y1 + X1)(Y2 + X2)", 2, int(2.53 * 2**29), 1, P_TOP),
    ("C = T1 (+/- 2dXY)", 1, int(1.27 * 2**29), 2, int(1.69 * 2**29)),
    ("X3 = E F'", 3, int(2.95 * 2**29), 1, int(4.21 * 2**29)),
    ("Y3 = G H'", 3, int(3.79 * 2**29), 1, int(2.53 * 2**29)),
    ("Z3 = F' G", 1, int(4.21 * 2**29), 3, int(3.79 * 2**29)),
    ("add A = (Y1 - X1)'(Y2 - X2)", 1, int(2.95 * 2**29), 3, int(2.95 * 2**29)),
]


@pytest.mark.parametrize("cls", CLASSES, ids=[c[0] for c in CLASSES])
def test_fe_mul_13x29_at_the_operand_classes_of_the_law(libs, cls):
    _, la, ta, lb, tb = cls
    rng = random.Random(sum(cls[0].encode()))
    n = 1 << 14
    a = operand_class(rng, n, la, ta)
    b = operand_class(rng, n, lb, tb)
    rng.shuffle(b)
    b[:64] = operand_class(rng, 64, lb, tb)       # extremes against extremes
    out, _ = run_both(libs, CID, "FE_MUL", arr([x + y for x, y in zip(a, b)]))
    check_class_m29(out)
    rinv = pow(R406, -1, P)
    for i in list(range(96)) + [rng.randrange(n) for _ in range(2000)]:
        assert value29(out[i]) % P == value29(a[i]) * value29(b[i]) * rinv % P, i


def test_twisted_edwards_law_13x29_matches_the_model(libs):
    import te_model as te

    c = m.BLS12_377_G1
    rng = random.Random(56)
    pts = m.random_points(c, 40, rng)
    timg = [te.sw_to_te(Q) for Q in pts]

    def ext_record(t, top_lazy=False):
        z = rng.randrange(1, P)
        X, Y = t
        out = []
        for v in (X * z % P, Y * z % P, z, X * Y % P * z % P):
            r = v * R406 % P
            out += limbs29(r + (P if top_lazy and r + P < P + (P >> 1) else 0))      # class M allows up to 1.5p
        return out

    def ext_to_affine(rec):
        X, Y, Z, T = (value29(rec[i * NW:(i + 1) * NW]) * pow(R406, -1, P) % P for i in range(4))
        assert Z != 0 and X * Y % P == Z * T % P
        zi = pow(Z, -1, P)
        return (X * zi % P, Y * zi % P)

    ident = (0, 1)
    cases = [(timg[0], timg[0]), (timg[1], te.te_neg(timg[1])), (ident, timg[2]), (timg[3], ident), (ident, ident)]
    while len(cases) < 320:
        cases.append((rng.choice(timg), rng.choice(timg)))
    recs = arr([ext_record(a, i % 3 == 0) + ext_record(b, i % 5 == 0) for i, (a, b) in enumerate(cases)])
    out, _ = run_both(libs, CID, "TE_ADD", recs)
    quad, _ = run_both(libs, CID, "TE_ADD_QUAD", recs, host_too=False)
    for i, (a, b) in enumerate(cases):
        exp = te.te_add(a, b)
        assert ext_to_affine(out[i].tolist()) == exp, ("TE_ADD", i)
        assert quad is None or ext_to_affine(quad[i].tolist()) == exp, ("TE_ADD_QUAD", i)
    check_class_m29(out.reshape(-1, NW))
    if quad is not None:
        check_class_m29(quad.reshape(-1, NW))
    out, _ = run_both(libs, CID, "TE_DBL", arr([ext_record(a) for a, _ in cases]))
    for i, (a, _) in enumerate(cases):
        assert ext_to_affine(out[i].tolist()) == te.te_add(a, a), ("TE_DBL", i)
    recs, recs_sw, negs = [], [], []
    for i, (a, b) in enumerate(cases):
        neg = rng.randrange(2)
        ymx, ypx, td = te.te_precomp(b)
        base = limbs29(ymx * R406 % P) + limbs29(ypx * R406 % P) + limbs29(td * R406 % P)
        base_sw = (limbs29(ypx * R406 % P) + limbs29(ymx * R406 % P) if neg else base[:2 * NW]) + base[2 * NW:]
        acc = ext_record(a, i % 2 == 0)
        recs.append(acc + base + [neg])
        recs_sw.append(acc + base_sw + [neg])
        negs.append(neg)
    out, _ = run_both(libs, CID, "TE_MADD", arr(recs))
    out_sw, _ = run_both(libs, CID, "TE_MADD_SWAPPED", arr(recs_sw))
    assert (out == out_sw).all()
    check_class_m29(out.reshape(-1, NW))
    for i, (a, b) in enumerate(cases):
        exp = te.te_add(a, te.te_neg(b) if negs[i] else b)
        assert ext_to_affine(out[i].tolist()) == exp, ("TE_MADD", i)


def test_law_13x29_at_the_largest_class_m_limbs(libs):
    """Every coordinate at the LARGEST limbs class M allows (2^29 - 1 below the top, the top limb of 1.5p) and canonical records at
    theirs: the values are not curve points -- only the column sums (host checker) and device == host limbs matter."""
    big = [MASK29] * (N29 - 1) + [(P + (P >> 1)) >> (B29 * (N29 - 1)), 0]
    rec = [MASK29] * (N29 - 1) + [P_TOP, 0]
    zero = [0] * NW
    rows = []
    for acc in ([big] * 4, [big, zero, big, zero], [zero, big, zero, big]):
        for base in ([rec] * 3, [rec, zero, rec], [zero, rec, zero]):
            for neg in (0, 1):
                rows.append(sum(acc, []) + sum(base, []) + [neg])
    run_both(libs, CID, "TE_MADD", arr(rows))
    run_both(libs, CID, "TE_MADD_SWAPPED", arr(rows))
    rows = [sum(a, []) + sum(b, []) for a in ([big] * 4, [big, zero, big, zero]) for b in ([big] * 4, [zero, big, zero, big])]
    run_both(libs, CID, "TE_ADD", arr(rows))
    run_both(libs, CID, "TE_ADD_QUAD", arr(rows), host_too=False)
