// curve.cuh -- short-Weierstrass (a = 0) group law in extended-Jacobian XYZZ coordinates over fp28.
//
// x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2.  Formulas are the public EFD ones the reference entries use
// (madd-2008-s, add-2008-s, dbl-2008-s-1, mdbl-2008-s-1): SPK ec/xyzz_t.hpp:97-170 (add),
// :178-249 (mixed add incl. doubling / infinity / negate), ML ec.cuh:495-600.  What is new here is
// the lazy-reduction schedule for the unsaturated radix-2^28 field: every value carries only a
// bound, and subtractions add a lifted multiple of p (F::BIASk_l) instead of borrowing.
//
// Stored-point invariants (what every function below both requires and re-establishes):
//   Xyzz.x, Xyzz.y : limbs < 2^28 + 16, value < 16p
//   Xyzz.zz, .zzz  : class M (normalized limbs, value < 2p);  infinity  <=>  zz == 0 (mod p)
//   Affine.x, .y   : class M (canonical < p when produced by the base-conversion kernel)
#pragma once
#include "fp28.cuh"

namespace msm {

struct Affine {
  Fe x, y;
};

struct Xyzz {
  Fe x, y, zz, zzz;
};

template <class F>
MSM_HD void xyzz_set_inf(Xyzz& r) {
  fe_zero(r.x);
  fe_zero(r.y);
  fe_zero(r.zz);
  fe_zero(r.zzz);
}

template <class F>
MSM_HD bool xyzz_is_inf(const Xyzz& a) {
  return fe_is_zero_M<F>(a.zz);
}

// r = (+/-) P as XYZZ with ZZ = ZZZ = 1.
template <class F>
MSM_HD void xyzz_from_affine(Xyzz& r, const Affine& p, bool negate) {
  r.x = p.x;
  Fe ny;
  fe_neg(ny, p.y, F::BIAS2_28);  // (0, 2p], limbs < 2^29
  fe_carry(ny);
  r.y = p.y;
  fe_cmov(r.y, ny, negate);
  fe_set(r.zz, F::ONE);
  fe_set(r.zzz, F::ONE);
}

// Shared tail of madd/add:  given P, R, PP = P^2 (already known non-zero mod p), the "first" point's
// U1 (=X1 for madd) and S1 (=Y1), produce X3, Y3 and return PPP for the ZZ/ZZZ updates.
//   X3 = R^2 - PPP - 2Q,  Y3 = R (Q - X3) - S1 PPP,  Q = U1 PP.
template <class F>
MSM_HD void add_tail(Fe& x3, Fe& y3, Fe& ppp, const Fe& P, Fe& R, const Fe& PP, const Fe& U1, const Fe& S1,
                     const Modulus<F>& md) {
  Fe q, r2, t, d, nppp;
  fe_mul<F>(ppp, P, PP, md);   // M
  fe_mul<F>(q, U1, PP, md);    // M
  fe_carry(R);                 // limbs < 2^28 + 16 (value unchanged): lets R share a reduction below
  fe_sqr<F>(r2, R, md);        // M
  fe_dbl(t, q);                // < 4p, limbs < 2^29
  fe_add(t, t, ppp);           // < 6p, limbs < 3*2^28
  fe_sub(x3, r2, t, F::BIAS8_30);  // (2p, 10p), limbs < 2^28 + 2^30 + 2^28
  fe_carry(x3);                // limbs < 2^28 + 16
  fe_sub(d, q, x3, F::BIAS16_29);  // (6p, 18p), limbs < 2^30
  fe_carry(d);                 // limbs < 2^28 + 16
  fe_neg(nppp, ppp, F::BIAS2_28);  // -PPP as (0, 2p], limbs < 2^29
  // Y3 = R*D - S1*PPP = R*D + S1*(-PPP): one reduction for both products; the result is class M
  fe_mul2<F>(y3, R, d, S1, nppp, md);
}

// acc = 2 * (x2, y2) from affine coordinates (mdbl-2008-s-1).  y2 may be a negated (lazy) value with
// limbs < 2^29.  A 2-torsion point (y = 0) yields ZZ = 0, i.e. infinity, with no special case.
template <class F>
MSM_HD void xyzz_dbl_affine(Xyzz& acc, const Fe& x2, const Fe& y2, const Modulus<F>& md) {
  Fe u, v, w, s, xx, m, mm, t, d, nw;
  fe_dbl(u, y2);               // limbs < 2^30, value <= 4p
  fe_sqr<F>(v, u, md);
  fe_mul<F>(w, u, v, md);
  fe_mul<F>(s, x2, v, md);
  fe_sqr<F>(xx, x2, md);
  fe_dbl(m, xx);
  fe_add(m, m, xx);            // 3*XX: < 6p, limbs < 3*2^28
  fe_carry(m);                 // limbs < 2^28 + 16
  fe_sqr<F>(mm, m, md);
  fe_dbl(t, s);                // < 4p, limbs < 2^29
  fe_sub(acc.x, mm, t, F::BIAS4_29);  // (0, 6p)
  fe_carry(acc.x);
  fe_sub(d, s, acc.x, F::BIAS8_29);   // (2p, 10p), limbs < 2^30
  fe_carry(d);
  fe_neg(nw, w, F::BIAS2_28);
  fe_mul2<F>(acc.y, m, d, y2, nw, md);  // M*(S - X3) - W*Y2   (y2 limbs < 2^29 also when negated)
  acc.zz = v;
  acc.zzz = w;
}

// acc = 2 * acc (dbl-2008-s-1).
template <class F>
MSM_HD void xyzz_dbl(Xyzz& acc, const Modulus<F>& md) {
  Fe u, v, w, s, xx, m, mm, t, d, nw, y1 = acc.y;
  fe_dbl(u, acc.y);            // limbs < 2^29 + 32, value < 32p
  fe_sqr<F>(v, u, md);
  fe_mul<F>(w, u, v, md);
  fe_mul<F>(s, acc.x, v, md);
  fe_sqr<F>(xx, acc.x, md);
  fe_dbl(m, xx);
  fe_add(m, m, xx);
  fe_carry(m);
  fe_sqr<F>(mm, m, md);
  fe_dbl(t, s);
  fe_sub(acc.x, mm, t, F::BIAS4_29);
  fe_carry(acc.x);
  fe_sub(d, s, acc.x, F::BIAS8_29);
  fe_carry(d);
  fe_neg(nw, w, F::BIAS2_28);
  fe_mul2<F>(acc.y, m, d, y1, nw, md);
  fe_mul<F>(acc.zz, v, acc.zz, md);
  fe_mul<F>(acc.zzz, w, acc.zzz, md);
}

// acc += (+/-)(x2, y2)   (madd-2008-s; 8M + 2S on the common path).
// `acc_inf` lets the caller pass what it already knows (a fresh run starts at infinity) so the common
// first-element case costs no field work.  The affine point must not be infinity (filtered upstream:
// the digit kernel emits no entries for bases flagged infinite).
template <class F>
MSM_HD void xyzz_madd(Xyzz& acc, const Affine& p, bool negate, bool acc_inf, const Modulus<F>& md) {
  if (acc_inf || xyzz_is_inf<F>(acc)) {
    xyzz_from_affine<F>(acc, p, negate);
    return;
  }
  Fe y2, ny;
  fe_neg(ny, p.y, F::BIAS2_28);  // limbs < 2^29 + 2^28... (bias limb < 2^29) -> < 2^30
  y2 = p.y;
  fe_cmov(y2, ny, negate);
  Fe u2, s2, P, R, PP;
  fe_mul<F>(u2, p.x, acc.zz, md);
  fe_mul<F>(s2, y2, acc.zzz, md);
  fe_sub(P, u2, acc.x, F::BIAS16_29);  // (0, 18p), limbs < 2^30
  fe_sub(R, s2, acc.y, F::BIAS16_29);
  fe_sqr<F>(PP, P, md);
  if (fe_is_zero_M<F>(PP)) {
    // same x: either the same point (double) or its negative (infinity).
    Fe r2;
    fe_sqr<F>(r2, R, md);
    if (fe_is_zero_M<F>(r2)) {
      xyzz_dbl_affine<F>(acc, p.x, y2, md);
    } else {
      xyzz_set_inf<F>(acc);
    }
    return;
  }
  Fe x3, y3, ppp;
  add_tail<F>(x3, y3, ppp, P, R, PP, acc.x, acc.y, md);
  acc.x = x3;
  acc.y = y3;
  fe_mul<F>(acc.zz, acc.zz, PP, md);
  fe_mul<F>(acc.zzz, acc.zzz, ppp, md);
}

// acc += b   (add-2008-s; 12M + 2S).
template <class F>
MSM_HD void xyzz_add(Xyzz& acc, const Xyzz& b, const Modulus<F>& md) {
  if (xyzz_is_inf<F>(b)) return;
  if (xyzz_is_inf<F>(acc)) {
    acc = b;
    return;
  }
  Fe u1, u2, s1, s2, P, R, PP;
  fe_mul<F>(u1, acc.x, b.zz, md);
  fe_mul<F>(u2, b.x, acc.zz, md);
  fe_mul<F>(s1, acc.y, b.zzz, md);
  fe_mul<F>(s2, b.y, acc.zzz, md);
  fe_sub(P, u2, u1, F::BIAS2_28);  // (0, 4p), limbs < 3*2^28
  fe_sub(R, s2, s1, F::BIAS2_28);
  fe_sqr<F>(PP, P, md);
  if (fe_is_zero_M<F>(PP)) {
    Fe r2;
    fe_sqr<F>(r2, R, md);
    if (fe_is_zero_M<F>(r2)) {
      xyzz_dbl<F>(acc, md);
    } else {
      xyzz_set_inf<F>(acc);
    }
    return;
  }
  Fe x3, y3, ppp, t;
  add_tail<F>(x3, y3, ppp, P, R, PP, u1, s1, md);
  acc.x = x3;
  acc.y = y3;
  fe_mul<F>(t, acc.zz, b.zz, md);
  fe_mul<F>(acc.zz, t, PP, md);
  fe_mul<F>(t, acc.zzz, b.zzz, md);
  fe_mul<F>(acc.zzz, t, ppp, md);
}

}  // namespace msm
