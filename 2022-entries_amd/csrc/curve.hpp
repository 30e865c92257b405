// curve.hpp -- short-Weierstrass (a = 0) group law in extended-Jacobian XYZZ coordinates over fp28.
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
#include "fp28.hpp"

namespace msm {

// Points are generic over the coordinate element type: Fe for G1 (Fp), Fe2 for G2 (Fp2).
template <class T>
struct AffineT {
  T x, y;
};

template <class T>
struct XyzzT {
  T x, y, zz, zzz;
};

using Affine = AffineT<Fe>;
using Xyzz = XyzzT<Fe>;

// ---- coordinate-field policies -------------------------------------------------------------------------------
// Every curve function below is written against this interface; contracts (limb / value bounds) are those of fp28.hpp:
//   mul/sqr inputs: limbs < 2^30, values <= 18p (what the formulas produce);  outputs: class M per component.
//   mul2(r, a, b, c, d) = a*b + c*d with all limbs < 2^29; output limbs < 2^28 + 16, value < 4p per component.
template <class F>
struct FpEl {
  using Fld = F;
  using T = Fe;
  using Md = Modulus<F>;
  static MSM_HD void mul(T& r, const T& a, const T& b, const Md& md) { fe_mul<F>(r, a, b, md); }
  static MSM_HD void sqr(T& r, const T& a, const Md& md) { fe_sqr<F>(r, a, md); }
  static MSM_HD void mul2(T& r, const T& a, const T& b, const T& c, const T& d, const Md& md) { fe_mul2<F>(r, a, b, c, d, md); }
  // The *_c forms take operands whose limbs are already carried (see Fp2El, where that saves most of the bookkeeping);
  // over Fp the plain multiplier accepts lazy limbs anyway, so they are the same functions and prep() is a no-op.
  template <bool B_BIG>
  static MSM_HD void mul_c(T& r, const T& a, const T& b, const Md& md) { fe_mul<F>(r, a, b, md); }
  static MSM_HD void sqr_c(T& r, const T& a, const Md& md) { fe_sqr<F>(r, a, md); }
  static MSM_HD void prep(T&) {}
  // r = a*b - c*d, class M;  d class M
  static MSM_HD void mul_sub_c(T& r, const T& a, const T& b, const T& c, const T& d, const Md& md) {
    T nd;
    fe_neg(nd, d, F::BIAS2_28);   // -d as (0, 2p], limbs < 2^29
    fe_mul2<F>(r, a, b, c, nd, md);
  }
  static MSM_HD void add(T& r, const T& a, const T& b) { fe_add(r, a, b); }
  static MSM_HD void dbl(T& r, const T& a) { fe_dbl(r, a); }
  static MSM_HD void sub(T& r, const T& a, const T& b, const uint32_t (&bias)[NL]) { fe_sub(r, a, b, bias); }
  static MSM_HD void neg(T& r, const T& b, const uint32_t (&bias)[NL]) { fe_neg(r, b, bias); }
  static MSM_HD void carry(T& r) { fe_carry(r); }
  static MSM_HD bool is_zero_M(const T& a) { return fe_is_zero_M<F>(a); }
  static MSM_HD void cmov(T& r, const T& a, bool take) { fe_cmov(r, a, take); }
  static MSM_HD void set_one(T& r) { fe_set(r, F::ONE); }
  static MSM_HD void zero(T& r) { fe_zero(r); }
  // ABI images: 12 u32 words per coordinate (x*2^384 mod p, canonical)
  static constexpr int WORDS = 12;
  static constexpr int ACC_WAVES = 2;   // k_accumulate: <= 256 VGPRs, two waves per SIMD saturate VALU issue
  static constexpr bool PREFETCH_BASE = true;
#ifndef MSM_SW_ENTRY_Q
#define MSM_SW_ENTRY_Q 2
#endif
  static constexpr bool ITER_BARRIER = false;
  static constexpr int ENTRY_Q = MSM_SW_ENTRY_Q;   // k_accumulate_glds entry queue: half a sector per refill (155 + 8 VGPRs <= 168)
  static MSM_HD void from_abi(T& r, const uint32_t* w, const Md& md) { fe_from_abi<F>(r, w, md); }
  static MSM_HD void to_abi(uint32_t* w, const T& a, const Md& md) { fe_to_abi<F>(w, a, md); }
  static MSM_HD void reduce(T& r) { fe_reduce<F>(r); }
  // normal-form integers (arkworks CanonicalSerialize: little-endian, not Montgomery) <-> internal
  static MSM_HD void from_plain(T& r, const uint32_t* w, const Md& md) {
    Fe t, c;
    fe_from_words(t, w);
    fe_set(c, F::RR);
    fe_mul<F>(r, t, c, md);
  }
  static MSM_HD void to_plain(uint32_t* w, const T& a, const Md& md) {
    Fe t, one;
    fe_zero(one);
    one.v[0] = 1;
    fe_mul<F>(t, a, one, md);   // x*R * 1 * R^-1 = x
    fe_reduce<F>(t);
    fe_to_words(w, t);
  }
};

struct Fe2 {
  Fe c0, c1;
};

// Fp2 = Fp[u]/(u^2 - BETA) with a small negative BETA (-5 for BLS12-377: ARKC bls12_377/src/fields/fq2.rs:13).
// (a0 + a1 u)(b0 + b1 u) = (a0 b0 + a1 (BETA b1)) + (a0 b1 + a1 b0) u: two fused dual products, i.e. 4 limb products and
// 2 Montgomery reductions -- the same multiply count as Karatsuba (quadratic_extension.rs:641-652) but both components
// come out class M, which keeps every bound of the Fp schedule valid component-wise.
template <class F, int NEG_BETA>
struct Fp2El {
  using Fld = F;
  using T = Fe2;
  using Md = Modulus<F>;
  static MSM_HD void mul(T& r, const T& a, const T& b, const Md& md) {
    Fe a0 = a.c0, a1 = a.c1, b0 = b.c0, b1 = b.c1, t;
    fe_carry(a0);
    fe_carry(a1);
    fe_carry(b0);
    fe_weak_reduce<F>(b1);                 // < 3p, normalized: NEG_BETA * b1 stays below 16p
    Fe s;
#pragma unroll
    for (int i = 0; i < NL; i++) s.v[i] = b1.v[i] * (uint32_t)NEG_BETA;   // limbs < 5 * 2^28
    fe_neg(t, s, F::BIAS16_31);            // BETA*b1 as (p, 16p], limbs < 2^31 + 2^28
    fe_carry(t);
    Fe c0, c1;
    fe_mul2<F>(c0, a0, b0, a1, t, md);     // <= 18p*18p + 18p*16p < 2^10 p^2
    fe_mul2<F>(c1, a0, b1, a1, b0, md);
    r.c0 = c0;
    r.c1 = c1;
  }
  static MSM_HD void sqr(T& r, const T& a, const Md& md) { mul(r, a, a, md); }
  static MSM_HD void mul2(T& r, const T& a, const T& b, const T& c, const T& d, const Md& md) {
    T t1, t2;
    mul(t1, a, b, md);
    mul(t2, c, d, md);
    fe_add(r.c0, t1.c0, t2.c0);            // < 3p, limbs < 2^29: fine as a stored coordinate
    fe_add(r.c1, t1.c1, t2.c1);
    fe_carry(r.c0);
    fe_carry(r.c1);
  }
  // The same product for operands that are already CARRIED (limbs < 2^28 + 16), which is what the group law's hot path
  // holds anyway: no carries, no weak reduction.  The BETA factor moves to the a side, the sign to the b side:
  //   c0 = a0 b0 + (5 a1)(K p - b1),   K p = BIAS2_28 for a class-M b (strictly normalized, < 2p), BIAS32_29 (B_BIG) for a
  //   carried b of value <= 18p.  Column bound of the fused product: 14 (2^56 + 1.25 * 2^30 * 1.5 * 2^29 + 2^56) < 0.93 * 2^64;
  //   value: (18p 18p + 90p 32p) / 2^15 p + p < 2p -- class M as ever.
  template <bool B_BIG>
  static MSM_HD void mul_c(T& r, const T& a, const T& b, const Md& md) {
    Fe s, nb, c0, c1;
#pragma unroll
    for (int i = 0; i < NL; i++) s.v[i] = a.c1.v[i] * (uint32_t)NEG_BETA;
    if (B_BIG) fe_neg(nb, b.c1, F::BIAS32_29); else fe_neg(nb, b.c1, F::BIAS2_28);
    fe_mul2<F>(c0, a.c0, b.c0, s, nb, md);
    fe_mul2<F>(c1, a.c0, b.c1, a.c1, b.c0, md);
    r.c0 = c0;
    r.c1 = c1;
  }
  // (a0 + a1 u)^2 = (a0^2 + BETA a1^2) + (2 a0 a1) u: the u part is ONE product, not two.  a carried, value <= 18p.
  static MSM_HD void sqr_c(T& r, const T& a, const Md& md) {
    Fe s, nb, d, c0, c1;
#pragma unroll
    for (int i = 0; i < NL; i++) s.v[i] = a.c1.v[i] * (uint32_t)NEG_BETA;
    fe_neg(nb, a.c1, F::BIAS32_29);
    fe_mul2<F>(c0, a.c0, a.c0, s, nb, md);
    fe_dbl(d, a.c1);                       // limbs < 2^29 + 32
    fe_mul<F>(c1, a.c0, d, md);
    r.c0 = c0;
    r.c1 = c1;
  }
  static MSM_HD void prep(T& a) { fe_carry(a.c0); fe_carry(a.c1); }
  // r = a*b - c*d as a stored coordinate (limbs < 2^28 + 16, value < 4p);  a, b, c carried (b of value <= 18p), d class M
  static MSM_HD void mul_sub_c(T& r, const T& a, const T& b, const T& c, const T& d, const Md& md) {
    T t1, t2;
    mul_c<true>(t1, a, b, md);
    mul_c<false>(t2, c, d, md);
    fe_sub(r.c0, t1.c0, t2.c0, F::BIAS2_28);
    fe_sub(r.c1, t1.c1, t2.c1, F::BIAS2_28);
    fe_carry(r.c0);
    fe_carry(r.c1);
  }
  static MSM_HD void add(T& r, const T& a, const T& b) { fe_add(r.c0, a.c0, b.c0); fe_add(r.c1, a.c1, b.c1); }
  static MSM_HD void dbl(T& r, const T& a) { fe_dbl(r.c0, a.c0); fe_dbl(r.c1, a.c1); }
  static MSM_HD void sub(T& r, const T& a, const T& b, const uint32_t (&bias)[NL]) {
    fe_sub(r.c0, a.c0, b.c0, bias);
    fe_sub(r.c1, a.c1, b.c1, bias);
  }
  static MSM_HD void neg(T& r, const T& b, const uint32_t (&bias)[NL]) { fe_neg(r.c0, b.c0, bias); fe_neg(r.c1, b.c1, bias); }
  static MSM_HD void carry(T& r) { fe_carry(r.c0); fe_carry(r.c1); }
  static MSM_HD bool is_zero_M(const T& a) { return fe_is_zero_M<F>(a.c0) && fe_is_zero_M<F>(a.c1); }
  static MSM_HD void cmov(T& r, const T& a, bool take) { fe_cmov(r.c0, a.c0, take); fe_cmov(r.c1, a.c1, take); }
  static MSM_HD void set_one(T& r) { fe_set(r.c0, F::ONE); fe_zero(r.c1); }
  static MSM_HD void zero(T& r) { fe_zero(r.c0); fe_zero(r.c1); }
  // ABI images: c0 | c1, 12 u32 words each (arkworks QuadExtField { c0, c1 })
  static constexpr int WORDS = 24;
#ifndef MSM_G2_ACC_WAVES
#define MSM_G2_ACC_WAVES 1
#endif
  static constexpr int ACC_WAVES = MSM_G2_ACC_WAVES;   // an Fp2 XYZZ accumulator alone is 112 VGPRs: take the whole 512-entry file
  static constexpr bool PREFETCH_BASE = false;
#ifndef MSM_G2_ENTRY_Q
#define MSM_G2_ENTRY_Q 0
#endif
#ifndef MSM_G2_ITER_BARRIER
#define MSM_G2_ITER_BARRIER 0
#endif
  // k_accumulate_glds: the four waves of a block meet at a barrier every iteration.  The G2 addition is ~118 KB of straight-line
  // code against a 64-KB instruction cache shared by two CUs, and one wave per SIMD hides no fetch latency: waves that run the
  // same code at the same time share the fetches
  static constexpr bool ITER_BARRIER = MSM_G2_ITER_BARRIER != 0;
  static constexpr int ENTRY_Q = MSM_G2_ENTRY_Q;   // no registers to spare (256 VGPRs + AGPRs)
  static MSM_HD void from_abi(T& r, const uint32_t* w, const Md& md) {
    fe_from_abi<F>(r.c0, w, md);
    fe_from_abi<F>(r.c1, w + 12, md);
  }
  static MSM_HD void to_abi(uint32_t* w, const T& a, const Md& md) {
    fe_to_abi<F>(w, a.c0, md);
    fe_to_abi<F>(w + 12, a.c1, md);
  }
  static MSM_HD void reduce(T& r) { fe_reduce<F>(r.c0); fe_reduce<F>(r.c1); }
  static MSM_HD void from_plain(T& r, const uint32_t* w, const Md& md) {
    FpEl<F>::from_plain(r.c0, w, md);
    FpEl<F>::from_plain(r.c1, w + 12, md);
  }
  static MSM_HD void to_plain(uint32_t* w, const T& a, const Md& md) {
    FpEl<F>::to_plain(w, a.c0, md);
    FpEl<F>::to_plain(w + 12, a.c1, md);
  }
};

// ---- inversion in the coordinate field (host fold, and the device precompute/normalise kernels) ------------------
// limb i (radix 2^28) of p - 2, with the borrow carried through low limbs that are < 2 (BLS12-377 has p = 1 mod 2^28)
template <class F>
constexpr uint32_t pm2_limb(int i) {
  int64_t borrow = 2;
  uint32_t out = 0;
  for (int k = 0; k <= i; k++) {
    int64_t d = (int64_t)F::P[k] - borrow;
    if (d < 0) {
      d += (int64_t)1 << LB;
      borrow = 1;
    } else {
      borrow = 0;
    }
    out = (uint32_t)d;
  }
  return out;
}

// a^(p-2) by square-and-multiply over the bits of p-2 (class-M in, class-M out); a == 0 gives 0.
template <class F>
MSM_HD void fe_inv(Fe& r, const Fe& a, const Modulus<F>& md) {
  Fe acc;
  fe_set(acc, F::ONE);
  bool started = false;
  for (int i = NL - 1; i >= 0; i--) {   // (not unrolled on purpose: 14 x 28 square-and-multiply steps)
    const uint32_t e = pm2_limb<F>(i);
#pragma unroll 1
    for (int b = LB - 1; b >= 0; b--) {
      if (started) fe_sqr<F>(acc, acc, md);
      if ((e >> b) & 1) {
        if (started) {
          fe_mul<F>(acc, acc, a, md);
        } else {
          acc = a;
          started = true;
        }
      }
    }
  }
  r = acc;
}

template <class F>
MSM_HD void el_inv(Fe& r, const Fe& a, const Modulus<F>& md, FpEl<F>*) {
  fe_inv<F>(r, a, md);
}
// (a0 + a1 u)^-1 = (a0 - a1 u) / (a0^2 - BETA a1^2)   (quadratic_extension.rs:323)
template <class F, int NB>
MSM_HD void el_inv(Fe2& r, const Fe2& a, const Modulus<F>& md, Fp2El<F, NB>*) {
  Fe n0, n1, n, ni, t, one;
  fe_sqr<F>(n0, a.c0, md);
  fe_sqr<F>(n1, a.c1, md);
#pragma unroll
  for (int i = 0; i < NL; i++) n.v[i] = n0.v[i] + n1.v[i] * (uint32_t)NB;   // a0^2 + NB a1^2, < 9p, limbs < 6*2^28
  fe_carry(n);
  fe_set(one, F::ONE);
  fe_mul<F>(n, n, one, md);   // back to class M (n * R * R^-1)
  fe_inv<F>(ni, n, md);
  fe_mul<F>(r.c0, a.c0, ni, md);
  fe_mul<F>(t, a.c1, ni, md);
  fe_neg(r.c1, t, F::BIAS2_28);
  fe_carry(r.c1);
}

template <class E>
MSM_HD void xyzz_set_inf(XyzzT<typename E::T>& r) {
  E::zero(r.x);
  E::zero(r.y);
  E::zero(r.zz);
  E::zero(r.zzz);
}

template <class E>
MSM_HD bool xyzz_is_inf(const XyzzT<typename E::T>& a) {
  return E::is_zero_M(a.zz);
}

// r = (+/-) P as XYZZ with ZZ = ZZZ = 1.
template <class E>
MSM_HD void xyzz_from_affine(XyzzT<typename E::T>& r, const AffineT<typename E::T>& p, bool negate) {
  r.x = p.x;
  typename E::T ny;
  E::neg(ny, p.y, E::Fld::BIAS2_28);  // (0, 2p], limbs < 2^29
  E::carry(ny);
  r.y = p.y;
  E::cmov(r.y, ny, negate);
  E::set_one(r.zz);
  E::set_one(r.zzz);
}

// Tail of the full addition:  given P (carried), R, PP = P^2 (already known non-zero mod p), the first point's U1 and S1
// (class M), produce X3, Y3 and return PPP for the ZZ/ZZZ updates.
//   X3 = R^2 - PPP - 2Q,  Y3 = R (Q - X3) - S1 PPP,  Q = U1 PP.
template <class E>
MSM_HD void add_tail(typename E::T& x3, typename E::T& y3, typename E::T& ppp, const typename E::T& P, typename E::T& R, const typename E::T& PP, const typename E::T& U1, const typename E::T& S1,
                     const typename E::Md& md) {
  typename E::T q, r2, t, d;
  E::template mul_c<false>(ppp, P, PP, md);   // M
  E::template mul_c<false>(q, U1, PP, md);    // M
  E::carry(R);                 // limbs < 2^28 + 16 (value unchanged)
  E::sqr_c(r2, R, md);         // M
  E::dbl(t, q);                // < 4p, limbs < 2^29
  E::add(t, t, ppp);           // < 6p, limbs < 3*2^28
  E::sub(x3, r2, t, E::Fld::BIAS8_30);  // (2p, 10p), limbs < 2^28 + 2^30 + 2^28
  E::carry(x3);                // limbs < 2^28 + 16
  E::sub(d, q, x3, E::Fld::BIAS16_29);  // (6p, 18p), limbs < 2^30
  E::carry(d);                 // limbs < 2^28 + 16
  E::mul_sub_c(y3, R, d, S1, ppp, md);  // Y3 = R*D - S1*PPP (over Fp: one reduction for both products)
}

// acc = 2 * (x2, y2) from affine coordinates (mdbl-2008-s-1).  y2 may be a negated (lazy) value with
// limbs < 2^29.  A 2-torsion point (y = 0) yields ZZ = 0, i.e. infinity, with no special case.
template <class E>
MSM_HD void xyzz_dbl_affine(XyzzT<typename E::T>& acc, const typename E::T& x2, const typename E::T& y2, const typename E::Md& md) {
  typename E::T u, v, w, s, xx, m, mm, t, d, nw;
  E::dbl(u, y2);               // limbs < 2^30, value <= 4p
  E::sqr(v, u, md);
  E::mul(w, u, v, md);
  E::mul(s, x2, v, md);
  E::sqr(xx, x2, md);
  E::dbl(m, xx);
  E::add(m, m, xx);            // 3*XX: < 6p, limbs < 3*2^28
  E::carry(m);                 // limbs < 2^28 + 16
  E::sqr(mm, m, md);
  E::dbl(t, s);                // < 4p, limbs < 2^29
  E::sub(acc.x, mm, t, E::Fld::BIAS4_29);  // (0, 6p)
  E::carry(acc.x);
  E::sub(d, s, acc.x, E::Fld::BIAS8_29);   // (2p, 10p), limbs < 2^30
  E::carry(d);
  E::neg(nw, w, E::Fld::BIAS2_28);
  E::mul2(acc.y, m, d, y2, nw, md);  // M*(S - X3) - W*Y2   (y2 limbs < 2^29 also when negated)
  acc.zz = v;
  acc.zzz = w;
}

// acc = 2 * acc (dbl-2008-s-1).
template <class E>
MSM_HD void xyzz_dbl(XyzzT<typename E::T>& acc, const typename E::Md& md) {
  typename E::T u, v, w, s, xx, m, mm, t, d, y1 = acc.y;
  E::dbl(u, acc.y);            // limbs < 2^29 + 32, value < 32p
  E::prep(u);
  E::sqr_c(v, u, md);
  E::template mul_c<false>(w, u, v, md);
  E::template mul_c<false>(s, acc.x, v, md);
  E::sqr_c(xx, acc.x, md);
  E::dbl(m, xx);
  E::add(m, m, xx);            // 3*XX: < 6p, limbs < 3*2^28
  E::carry(m);
  E::sqr_c(mm, m, md);
  E::dbl(t, s);
  E::sub(acc.x, mm, t, E::Fld::BIAS4_29);
  E::carry(acc.x);
  E::sub(d, s, acc.x, E::Fld::BIAS8_29);   // (2p, 10p), limbs < 2^30
  E::carry(d);
  E::mul_sub_c(acc.y, m, d, y1, w, md);    // M*(S - X3) - Y1*W
  E::template mul_c<false>(acc.zz, v, acc.zz, md);
  E::template mul_c<false>(acc.zzz, w, acc.zzz, md);
}

// The rare half of a mixed addition: acc and (+/-)(x2, y2) have the same x, so the sum is either the double of the point or
// infinity (SPK ec/xyzz_t.hpp:214-236 handles the same two cases inline).
template <class E>
MSM_HD void xyzz_madd_same_x(XyzzT<typename E::T>& acc, const AffineT<typename E::T>& p, bool negate, const typename E::Md& md) {
  typename E::T y2, ny, s2, R, r2;
  E::neg(ny, p.y, E::Fld::BIAS2_28);
  y2 = p.y;
  E::cmov(y2, ny, negate);
  E::mul(s2, y2, acc.zzz, md);
  E::sub(R, s2, acc.y, E::Fld::BIAS16_29);
  E::sqr(r2, R, md);
  if (E::is_zero_M(r2)) {
    xyzz_dbl_affine<E>(acc, p.x, y2, md);
  } else {
    xyzz_set_inf<E>(acc);
  }
}

// acc += (+/-)(x2, y2)   (madd-2008-s; 8M + 2S), the common path only: returns true -- with acc untouched -- when the two
// points share their x coordinate and xyzz_madd_same_x must finish the job.  The split exists for the accumulate kernel:
// the rare branch would otherwise keep both base coordinates alive across the whole addition (56 VGPRs on G2); the kernel
// re-reads the base from memory instead when it happens.
// `acc_inf` lets the caller pass what it already knows (a fresh run starts at infinity) so the common
// first-element case costs no field work.  The affine point must not be infinity (filtered upstream:
// the digit kernel emits no entries for bases flagged infinite).
// The order of the operations is chosen for register pressure (an Fp2 element is 28 VGPRs): every old coordinate dies as
// early as it can.
template <class E>
MSM_HD bool xyzz_madd_common(XyzzT<typename E::T>& acc, const AffineT<typename E::T>& p, bool negate, bool acc_inf, const typename E::Md& md) {
  if (acc_inf || xyzz_is_inf<E>(acc)) {
    xyzz_from_affine<E>(acc, p, negate);
    return false;
  }
  // Operand classes: p.x, ZZ, ZZZ, PP, PPP are class M; X1, Y1 are stored coordinates (carried limbs); P, y2, R, D are
  // brought to carried limbs by prep()/carry() -- which is all the *_c multipliers ask for.
  typename E::T u2, P, PP;
  E::template mul_c<false>(u2, p.x, acc.zz, md);
  E::sub(P, u2, acc.x, E::Fld::BIAS16_29);  // (0, 18p), limbs < 2^30
  E::prep(P);
  E::sqr_c(PP, P, md);
  if (E::is_zero_M(PP)) return true;
  typename E::T y2, ny, s2, R;
  E::neg(ny, p.y, E::Fld::BIAS2_28);         // (0, 2p], limbs < 2^29
  y2 = p.y;
  E::cmov(y2, ny, negate);
  E::prep(y2);
  E::template mul_c<false>(s2, y2, acc.zzz, md);
  E::sub(R, s2, acc.y, E::Fld::BIAS16_29);
  E::carry(R);                               // limbs < 2^28 + 16 (value unchanged, <= 18p)
  E::template mul_c<false>(acc.zz, acc.zz, PP, md);     // old ZZ dies
  typename E::T ppp, q;
  E::template mul_c<false>(ppp, P, PP, md);             // P dies
  E::template mul_c<false>(q, acc.x, PP, md);           // old X and PP die
  E::template mul_c<false>(acc.zzz, acc.zzz, ppp, md);  // old ZZZ dies
  // X3 = R^2 - PPP - 2Q,  Y3 = R (Q - X3) - Y1 PPP
  typename E::T r2, t, d;
  E::sqr_c(r2, R, md);
  E::dbl(t, q);                              // < 4p, limbs < 2^29
  E::add(t, t, ppp);                         // < 6p, limbs < 3*2^28
  E::sub(acc.x, r2, t, E::Fld::BIAS8_30);    // (2p, 10p), limbs < 2^28 + 2^30 + 2^28
  E::carry(acc.x);                           // limbs < 2^28 + 16
  E::sub(d, q, acc.x, E::Fld::BIAS16_29);    // (6p, 18p), limbs < 2^30
  E::carry(d);                               // limbs < 2^28 + 16
  typename E::T y3;
  E::mul_sub_c(y3, R, d, acc.y, ppp, md);    // over Fp: one reduction for both products
  acc.y = y3;
  return false;
}

// acc += (+/-)(x2, y2), every case.
template <class E>
MSM_HD void xyzz_madd(XyzzT<typename E::T>& acc, const AffineT<typename E::T>& p, bool negate, bool acc_inf, const typename E::Md& md) {
  if (xyzz_madd_common<E>(acc, p, negate, acc_inf, md)) xyzz_madd_same_x<E>(acc, p, negate, md);
}

// acc += b   (add-2008-s; 12M + 2S).
template <class E>
MSM_HD void xyzz_add(XyzzT<typename E::T>& acc, const XyzzT<typename E::T>& b, const typename E::Md& md) {
  if (xyzz_is_inf<E>(b)) return;
  if (xyzz_is_inf<E>(acc)) {
    acc = b;
    return;
  }
  typename E::T u1, u2, s1, s2, P, R, PP;
  E::template mul_c<false>(u1, acc.x, b.zz, md);
  E::template mul_c<false>(u2, b.x, acc.zz, md);
  E::template mul_c<false>(s1, acc.y, b.zzz, md);
  E::template mul_c<false>(s2, b.y, acc.zzz, md);
  E::sub(P, u2, u1, E::Fld::BIAS2_28);  // (0, 4p), limbs < 3*2^28
  E::sub(R, s2, s1, E::Fld::BIAS2_28);
  E::prep(P);
  E::sqr_c(PP, P, md);
  if (E::is_zero_M(PP)) {
    typename E::T r2;
    E::sqr(r2, R, md);
    if (E::is_zero_M(r2)) {
      xyzz_dbl<E>(acc, md);
    } else {
      xyzz_set_inf<E>(acc);
    }
    return;
  }
  typename E::T x3, y3, ppp, t;
  add_tail<E>(x3, y3, ppp, P, R, PP, u1, s1, md);
  acc.x = x3;
  acc.y = y3;
  E::template mul_c<false>(t, acc.zz, b.zz, md);
  E::template mul_c<false>(acc.zz, t, PP, md);
  E::template mul_c<false>(t, acc.zzz, b.zzz, md);
  E::template mul_c<false>(acc.zzz, t, ppp, md);
}

#if defined(__HIPCC__)
// ---- one full addition by the four lanes of a quad -------------------------------------------------------------------------
// The latency form of xyzz_add for the kernels that are chains of DEPENDENT additions on an idle chip (fragment merge and
// scan reduction of small and medium inputs; msm_kernels.hpp).  Lane q of a quad owns coordinate q of both operands and of the
// result -- 0 X, 1 Y, 2 ZZ, 3 ZZZ, the order they lie in memory -- and the 12M + 2S of add-2008-s become four multiplications
// in a row, operands exchanged by DPP quad permutes:
//   1   U1 = X1 ZZ2 | S1 = Y1 ZZZ2 | U2 = X2 ZZ1 | S2 = Y2 ZZZ1          (own a times the b of lane q ^ 2)
//   2   PP = P^2    | RR = R^2     | ZZ1 ZZ2     | ZZZ1 ZZZ2             (P = U2 - U1, R = S2 - S1)
//   3   PPP = P PP  | Q = U1 PP    | ZZ3 = (ZZ1 ZZ2) PP | --
//   4   S1 PPP      | R (Q - X3)   | --          | ZZZ3 = (ZZZ1 ZZZ2) PPP;   X3 = RR - PPP - 2Q,  Y3 = R (Q - X3) - S1 PPP
// Same bounds as xyzz_add / add_tail.  Infinity operands and the same-x cases (doubling, cancellation) are decided
// quad-uniformly; the doubling -- rare -- gathers the point and runs the one-lane formula on every lane.
template <int CTRL, int N = NL>
__device__ __forceinline__ void fe_quad_perm(Fe& r, const Fe& a) {
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.v[i], CTRL, 0xf, 0xf, true);
#pragma unroll
  for (int i = N; i < NL; i++) r.v[i] = 0;
}
template <int CTRL>
__device__ __forceinline__ bool quad_flag(bool z) {   // one lane's flag to the whole quad (CTRL = 0x00 / 0x55 / 0xAA / 0xFF: lane 0..3)
  return __builtin_amdgcn_update_dpp(0, (int)z, CTRL, 0xf, 0xf, true) != 0;
}
template <int N = NL>
__device__ __forceinline__ void fe_select(Fe& r, const Fe& a, const Fe& b, bool take_b) {   // r = take_b ? b : a
  const LaneMask m = lane_mask(take_b);
  r = a;
  fe_cmov<N>(r, b, m);
}
// the same two helpers for Fp2 coordinates
template <int CTRL>
__device__ __forceinline__ void fe_quad_perm(Fe2& r, const Fe2& a) {
  fe_quad_perm<CTRL>(r.c0, a.c0);
  fe_quad_perm<CTRL>(r.c1, a.c1);
}
__device__ __forceinline__ void fe_select(Fe2& r, const Fe2& a, const Fe2& b, bool take_b) {
  fe_select(r.c0, a.c0, b.c0, take_b);
  fe_select(r.c1, a.c1, b.c1, take_b);
}

// E = FpEl<F> (G1) or Fp2El<F, NB> (G2): only the coordinate type and its multiplier differ.
template <class E>
__device__ __forceinline__ void xyzz_add_quad(typename E::T& a, const typename E::T& b, uint32_t q, const typename E::Md& md) {
  using T = typename E::T;
  using F = typename E::Fld;
  // infinity <=> ZZ == 0: lane 2 knows
  if (quad_flag<0xAA>(E::is_zero_M(b))) return;
  if (quad_flag<0xAA>(E::is_zero_M(a))) {
    a = b;
    return;
  }
  T pb, r1, d1, r2, r3, r4, u, v;
  fe_quad_perm<0x4E>(pb, b);                       // lanes 0 <-> 2, 1 <-> 3
  E::mul(r1, a, pb, md);                           // U1 | S1 | U2 | S2
  fe_quad_perm<0x4E>(d1, r1);
  E::sub(d1, d1, r1, F::BIAS2_28);                 // lane 0: P, lane 1: R  -- (0, 4p), limbs < 3*2^28
  E::carry(d1);                                    // limbs < 2^28 + 16
  fe_select(u, a, d1, q < 2);
  fe_select(v, b, d1, q < 2);
  E::mul(r2, u, v, md);                            // PP | RR | ZZ1 ZZ2 | ZZZ1 ZZZ2
  if (quad_flag<0x00>(E::is_zero_M(r2))) {
    // same x: the double of the point, or infinity
    if (quad_flag<0x55>(E::is_zero_M(r2))) {
      XyzzT<T> pt;
      fe_quad_perm<0x00>(pt.x, a);
      fe_quad_perm<0x55>(pt.y, a);
      fe_quad_perm<0xAA>(pt.zz, a);
      fe_quad_perm<0xFF>(pt.zzz, a);
      xyzz_dbl<E>(pt, md);
      fe_select(u, pt.x, pt.y, q == 1);
      fe_select(v, pt.zz, pt.zzz, q == 3);
      fe_select(a, u, v, q >= 2);
    } else {
      E::zero(a);
    }
    return;
  }
  T PPb, U1b;
  fe_quad_perm<0x00>(PPb, r2);
  fe_quad_perm<0x00>(U1b, r1);
  fe_select(u, r2, d1, q == 0);
  fe_select(u, u, U1b, q == 1);
  E::mul(r3, u, PPb, md);                          // PPP | Q | ZZ3 | (unused)
  T PPPb, Qb, RRb, S1b, t, x3, d;
  fe_quad_perm<0x00>(PPPb, r3);
  fe_quad_perm<0x55>(Qb, r3);
  fe_quad_perm<0x55>(RRb, r2);
  fe_quad_perm<0x55>(S1b, r1);
  E::dbl(t, Qb);                                   // < 4p, limbs < 2^29
  E::add(t, t, PPPb);                              // < 6p, limbs < 3*2^28
  E::sub(x3, RRb, t, F::BIAS8_30);                 // (2p, 10p), limbs < 2^28 + 2^30 + 2^28
  E::carry(x3);                                    // limbs < 2^28 + 16
  E::sub(d, Qb, x3, F::BIAS16_29);                 // (6p, 18p), limbs < 2^30
  E::carry(d);
  fe_select(u, r2, S1b, q == 0);                   // lane 0: S1 PPP;  lane 1: R d;  lane 3: (ZZZ1 ZZZ2) PPP
  fe_select(u, u, d1, q == 1);
  fe_select(v, PPPb, d, q == 1);
  E::mul(r4, u, v, md);
  T sp, y3;
  fe_quad_perm<0x00>(sp, r4);                      // S1 PPP
  E::sub(y3, r4, sp, F::BIAS2_28);                 // lane 1: R d - S1 PPP, (0, 4p)
  E::carry(y3);
  fe_select(u, x3, y3, q == 1);
  fe_select(v, r3, r4, q == 3);
  fe_select(a, u, v, q >= 2);
}
#endif

}  // namespace msm
