#!/usr/bin/env python3
"""Worst-case column sums of the 13 x 29 twisted-Edwards law (csrc/te.hpp over Bls12_377_Fq29, csrc/fp28.hpp fe_mul).

fe_mul gathers every a_i*b_j and m_i*p_j of a column in ONE 64-bit accumulator.  For 14 x 28 limbs that is safe with two bits of
lazy headroom per operand; 13 x 29 has one bit less, so this script carries an UPPER BOUND PER LIMB through every step of
te_madd / te_add / te_add_quad exactly as the code performs them (biased subtraction, limb-wise sums, parallel carry passes that do
not shorten the top limb) and evaluates, for every product, the largest value any column can reach:

    col_k <= sum_{i+j=k} A_i B_j  +  (2^29 - 1) * sum_{i+j=k, j>=1} p_j  +  (2^29 - 1)        [m_k * p_0, p_0 = 1]
             + floor(col_{k-1} / 2^29)

with the exact limbs of p.  It also checks that no biased subtraction can underflow a limb or overflow 32 bits.
tests/test_limbs29_host.py runs it (all margins must be positive); the same sums are checked dynamically, on every product, by the
MSM_CHECK build of the same templates (libmsm_hosttest.so).

    python tools/limb_bounds29.py            # prints the table that profiles/r06_ab_limbs29.txt quotes
"""
import sys

P = 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001
N, B, NRED = 13, 29, 14
MASK = (1 << B) - 1


def limbs(x):
    return [(x >> (B * i)) & MASK for i in range(N - 1)] + [x >> (B * (N - 1))]


PL = limbs(P)


def bias(k, lift=B):
    m = limbs(k * P)
    hi = 1 << (lift - B)
    out = [m[0] + (1 << lift)] + [m[i] + (1 << lift) - hi for i in range(1, N - 1)] + [m[N - 1] - hi]
    assert sum(v << (B * i) for i, v in enumerate(out)) == k * P and all(0 <= v < 1 << 32 for v in out)
    return out


BIAS2 = bias(2)


class V:
    """upper bounds: per limb, and of the value in units of p (a float, for the record only)."""

    def __init__(self, lim, val, name=""):
        self.l, self.val, self.name = list(lim), val, name

    def __repr__(self):
        return "%s: limbs <= %.3f, top <= %.3f (x 2^29), value < %.2fp" % (self.name, max(self.l[:-1]) / 2**B, self.l[-1] / 2**B, self.val)


def class_m(name, val=1.5):
    """a multiplier's output / a stored coordinate: limbs 0..11 < 2^29, value < val * p"""
    return V([MASK] * (N - 1) + [int(val * P) >> (B * (N - 1))], val, name)


def canonical(name):
    return V([MASK] * (N - 1) + [PL[N - 1]], 1.0, name)


RESULTS = []


def sub(name, a, b, bs=BIAS2, k=2):
    for i in range(N):
        assert bs[i] >= b.l[i], ("bias underflow", name, i)
        assert a.l[i] + bs[i] < 1 << 32, ("u32 overflow", name, i)
    assert b.val <= k, (name, "value of the subtrahend exceeds the bias")
    return V([a.l[i] + bs[i] for i in range(N)], a.val + k, name)


def neg_or_keep(name, b, bs=BIAS2, k=2):
    for i in range(N):
        assert bs[i] >= b.l[i], ("bias underflow", name, i)
    return V([max(b.l[i], bs[i]) for i in range(N)], k, name)


def add(name, a, b):
    out = V([a.l[i] + b.l[i] for i in range(N)], a.val + b.val, name)
    assert all(x < 1 << 32 for x in out.l), name
    return out


def dbl(name, a):
    return add(name, a, a)


def carry(name, a):
    c = [x >> B for x in a.l]
    out = [min(a.l[0], MASK)] + [MASK + c[i - 1] for i in range(1, N - 1)] + [a.l[N - 1] + c[N - 2]]
    assert all(x < 1 << 32 for x in out)
    return V(out, a.val, name)


def pick(name, *vs):
    """a select between alternatives (fe_select / lanes of a quad): the limb-wise maximum"""
    return V([max(v.l[i] for v in vs) for i in range(N)], max(v.val for v in vs), name)


def mul(name, a, b):
    col, worst, worst_k = 0, 0, 0
    for k in range(NRED + N - 1):
        col >>= B
        for i in range(N):
            j = k - i
            if 0 <= j < N:
                col += a.l[i] * b.l[j]
        for i in range(NRED):
            j = k - i
            if 1 <= j < N:
                col += MASK * PL[j]
        if k < NRED:
            col += MASK  # m_k * p_0
        if col > worst:
            worst, worst_k = col, k
    ok = worst < 1 << 64
    RESULTS.append((name, a, b, worst, worst_k, ok))
    assert a.val * b.val <= 1024, name
    return class_m(name)


def tail(A, Bv, C, D, pre=""):
    e = sub(pre + "E = B - A", Bv, A)
    f = sub(pre + "F = D - C", D, C)
    g = add(pre + "G = D + C", D, C)
    h = add(pre + "H = B + A", Bv, A)
    f = carry(pre + "F carried", f)
    h = carry(pre + "H carried", h)
    mul(pre + "X3 = E F", e, f)
    mul(pre + "Y3 = G H", g, h)
    mul(pre + "T3 = E H", e, h)
    mul(pre + "Z3 = F G", f, g)
    return e, f, g, h


def main():
    X, Y, Z, T = (class_m(n) for n in "XYZT")
    ymx, ypx, td = canonical("Y-X (record)"), canonical("Y+X (record)"), canonical("2dXY (record)")
    # te_madd
    td2 = neg_or_keep("+/- 2dXY", td)
    a1 = sub("Y1 - X1", Y, X)
    b1 = add("Y1 + X1", Y, X)
    A = mul("madd  A = (Y1 - X1)(Y2 - X2)", a1, ymx)
    Bv = mul("madd  B = (Y1 + X1)(Y2 + X2)", b1, ypx)
    C = mul("madd  C = T1 (2d X2 Y2)", T, td2)
    D = dbl("D = 2 Z1", Z)
    tail(A, Bv, C, D, "tail  ")
    # te_add
    a1c = carry("Y1 - X1 carried", a1)
    b1c = carry("Y1 + X1 carried", b1)
    mul("add   kT2 = T2 k", T, canonical("2d"))
    mul("add   Z1 Z2", Z, Z)
    mul("add   A = (Y1 - X1)'(Y2 - X2)", a1c, a1)
    mul("add   B = (Y1 + X1)'(Y2 + X2)", b1c, b1)
    mul("add   C = T1 (k T2)", T, class_m("kT2"))
    # te_add_quad: u = carry(select(Y+X, Y-X)), then lanes 2, 3 take Z1 | k; v = select(Y2+X2, Y2-X2, Z2 | T2)
    u = pick("quad u", carry("u carried", pick("u", a1, b1)), Z, canonical("2d"))
    v = pick("quad v", a1, b1, Z)
    mul("quad  step 1 (A | B | Z1 Z2 | k T2)", u, v)
    mul("quad  step 2 (a r1)", class_m("own"), class_m("r1"))
    tail(A, Bv, C, D, "quad  step 3  ")   # lane q multiplies ONE of the four pairs of te_tail (the selects only route them)
    # the re-radixing products (fe_28_to_29 / fe_29_to_28 run fe_mul on a normalized value < 2^380 and a canonical constant)
    mul("conv  regrouped value x constant", V([MASK] * (N - 1) + [(1 << 32) - 1], 9.5, "regrouped"), canonical("FROM28"))

    print("13 x 29 limbs, 14 Montgomery steps: worst-case column sums (limb bounds in units of 2^29)")
    print("%-40s %-22s %-22s %10s  %s" % ("product", "a: limbs / top", "b: limbs / top", "max col", "margin to 2^64"))
    bad = 0
    for name, a, b, worst, k, ok in RESULTS:
        print("%-40s %6.3f / %-13.3f %6.3f / %-13.3f %7.3f * 2^58 (k=%2d)  %5.1f %%%s" %
              (name, max(a.l[:-1]) / 2**B, a.l[-1] / 2**B, max(b.l[:-1]) / 2**B, b.l[-1] / 2**B, worst / 2**58, k, 100.0 * ((1 << 64) - worst) / (1 << 64),
               "" if ok else "   OVERFLOW"))
        bad += not ok
    mp = max(sum(MASK * PL[k - i] for i in range(NRED) if 1 <= k - i < N) for k in range(NRED + N - 1))
    print("largest m*p share of a column: %.3f * 2^58 (13 * 2^58 would be the bound without the limbs of p)" % (mp / 2**58))
    # what does NOT work, for the record: the same law without the carries, and a 13-step reduction
    print()
    print("without the two carry passes of te_tail:  F = D - C + bias has limbs < 4 * 2^29, H = B + A < 2 * 2^29, E, G < 3 * 2^29")
    for nm, la, lb in (("E F", 3, 4), ("G H", 3, 2), ("E H", 3, 2), ("F G", 4, 3)):
        print("    %s: 13 * %d * %d = %d * 2^58 + m*p %.1f * 2^58 %s 64 * 2^58" % (nm, la, lb, 13 * la * lb, mp / 2**58, "<" if 13 * la * lb + mp / 2**58 < 64 else ">"))
    r13 = P / 2.0**377
    print("13 reduction steps (R = 2^377): p/R = %.3f, so a product of operands < a p and < b p comes out < (1 + %.3f a b) p:" % (r13, r13))
    v = 1.0
    for it in range(4):
        A_, B_, C_ = 1 + r13 * (v + 2), 1 + r13 * 2 * v, 1 + r13 * v * 2
        e_, f_, g_, h_ = B_ + 2 + 0, 2 * v + 2, 2 * v + C_, A_ + B_   # biases of 2p where the subtrahend is < 2p (it is not, from the 2nd round on)
        v = 1 + r13 * max(e_ * f_, g_ * h_, e_ * h_, f_ * g_)
        print("    after %d mixed addition(s): coordinates < %.1f p" % (it + 1, v))
    print("  -> no fixed point: lazy reduction needs R >= 2^(377 + 10), i.e. the 14th step (168 instead of 156 m*p multiply-adds).")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
