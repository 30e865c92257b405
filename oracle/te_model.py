"""TEST INFRASTRUCTURE -- big-int model of the twisted-Edwards image of BLS12-377 G1 used by the accelerated path.

BLS12-377 G1 (y^2 = x^3 + 1) has a rational 2-torsion point (-1, 0) and 3 is a square mod p, so the curve is birationally
equivalent to a Montgomery and then to a twisted Edwards curve -X^2 + Y^2 = 1 + d X^2 Y^2 (the trick of the Trapdoor-Tech
entry, P1A/Trapdoor-Tech/msm_opt.md; extended coordinates and the 7M mixed addition of P1A/Trapdoor-Tech/sppark/ec/exte_t.hpp).
The constants are derived here from first principles (nothing is copied):

    alpha = -1, s = 1/sqrt(3):   (x, y)  ->  Montgomery (u, v) = (s (x + 1), s y)      B v^2 = u^3 + A u^2 + u, A = -3 s, B = s
    a_te = (A + 2)/B, d_te = (A - 2)/B:   (u, v)  ->  (u / v, (u - 1)/(u + 1))           a_te X^2 + Y^2 = 1 + d_te X^2 Y^2
    f = sqrt(-a_te):   X' = f X                                                        -X'^2 + Y^2 = 1 + d X'^2 Y^2,  d = -d_te/a_te

d is a SQUARE mod p, so the addition law is complete only on odd-order subgroups; the engine detects Z3 = 0 and falls back.
"""
from pymodel import BLS12_377_G1 as C

P = C.p


def _sqrt(a):
    a %= P
    if a == 0:
        return 0
    assert pow(a, (P - 1) // 2, P) == 1, "not a square"
    q, s = P - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (P - 1) // 2, P) != P - 1:
        z += 1
    m, c, t, r = s, pow(z, q, P), pow(a, q, P), pow(a, (q + 1) // 2, P)
    while t != 1:
        i, tt = 0, t
        while tt != 1:
            tt = tt * tt % P
            i += 1
        b = pow(c, 1 << (m - i - 1), P)
        m, c = i, b * b % P
        t, r = t * c % P, r * b % P
    return min(r, P - r)          # canonical choice of the root: the smaller one


SQRT3 = _sqrt(3)
S = pow(SQRT3, -1, P)
A_M = (-3 * S) % P
B_M = S
A_TE = (A_M + 2) * pow(B_M, -1, P) % P
D_TE = (A_M - 2) * pow(B_M, -1, P) % P
FSC = _sqrt(-A_TE)
D = (-D_TE * pow(A_TE, -1, P)) % P
K2D = 2 * D % P


def on_te(pt):
    x, y = pt
    return (-x * x + y * y - 1 - D * x * x % P * y * y) % P == 0


def sw_to_te(pt):
    """SW affine (x, y) -> TE affine (X, Y); None for the points the map is not defined on (y = 0, or u = -1)."""
    if pt is None:
        return (0, 1)
    x, y = pt
    u, v = S * (x + 1) % P, S * y % P
    if v == 0 or (u + 1) % P == 0:
        return None
    return (FSC * u % P * pow(v, -1, P) % P, (u - 1) * pow(u + 1, -1, P) % P)


def te_to_sw(pt):
    X, Y = pt
    if X == 0:
        return None if Y == 1 else (P - 1, 0)          # identity -> infinity; (0, -1) -> the 2-torsion point (-1, 0)
    u = (1 + Y) * pow(1 - Y, -1, P) % P
    v = FSC * u % P * pow(X, -1, P) % P
    return ((u * SQRT3 - 1) % P, v * SQRT3 % P)


def te_add(p1, p2):
    """Affine unified addition; returns None when a denominator vanishes (possible only off the odd-order subgroup)."""
    x1, y1 = p1
    x2, y2 = p2
    t = D * x1 % P * x2 % P * y1 % P * y2 % P
    if (1 + t) % P == 0 or (1 - t) % P == 0:
        return None
    x3 = (x1 * y2 + y1 * x2) * pow(1 + t, -1, P) % P
    y3 = (y1 * y2 + x1 * x2) * pow(1 - t, -1, P) % P     # a = -1:  y1 y2 - a x1 x2
    return (x3, y3)


def te_neg(pt):
    return ((-pt[0]) % P, pt[1])


def te_precomp(pt):
    """The device base record: (Y - X, Y + X, 2 d X Y)."""
    X, Y = pt
    return ((Y - X) % P, (Y + X) % P, K2D * X % P * Y % P)


def exceptional_points():
    """SW points the birational map is undefined on: the three 2-torsion points and the (up to two) points with u = -1."""
    out = []
    r3m = _sqrt(-3)
    for x in (P - 1, (1 + r3m) * pow(2, -1, P) % P, (1 - r3m) * pow(2, -1, P) % P):
        assert (x * x * x + 1) % P == 0
        out.append((x, 0))
    x = (-SQRT3 - 1) % P            # u = s (x + 1) = -1
    y2 = (x * x * x + 1) % P
    if pow(y2, (P - 1) // 2, P) == 1:
        y = _sqrt(y2)
        out += [(x, y), (x, P - y)]
    return out
