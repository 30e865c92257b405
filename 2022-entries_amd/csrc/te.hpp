// te.hpp -- twisted-Edwards image of BLS12-377 G1 in extended coordinates, over fp28.
//
// y^2 = x^3 + 1 over the BLS12-377 base field has the rational 2-torsion point (-1, 0) and 3 is a square, so the curve is
// birationally equivalent to  -X^2 + Y^2 = 1 + d X^2 Y^2  (the "curve isogeny" trick of the Trapdoor-Tech entry,
// P1A/Trapdoor-Tech/msm_opt.md; extended coordinates as in P1A/Trapdoor-Tech/sppark/ec/exte_t.hpp):
//     u = S (x + 1), v = S y  (S = 1/sqrt 3)        X = FSC u / v,  Y = (u - 1)/(u + 1)
// A mixed addition is 7 multiplications (EFD madd-2008-hwcd-3 with the 2d factor folded into the base record) against 8M + 2S
// for XYZZ, there is no doubling / infinity / cancellation branch (the law is unified and the identity (0, 1) is an ordinary
// point), and a full addition is 9M against 12M + 2S.  The constants are derived in oracle/te_model.py.
//
// Caveats, both handled by the engine rather than assumed away:
//   * five curve points have no image (the three 2-torsion points and the two points with u = -1): the base converter
//     counts them and a base set that contains one stays on the short-Weierstrass path;
//   * d is a SQUARE mod p, so the law is complete only on odd-order subgroups (every input of the harness lies in the
//     r-torsion).  Off the subgroup an addition can hit a vanishing denominator, which shows as Z3 = 0: every kernel
//     checks it after each addition and raises a flag, and the engine then repeats the run on the XYZZ path.
//
// Two limb shapes (fp28.hpp): the law below is written once over the field constants F and runs either on Bls12_377_Fq (14 x 28,
// R = 2^392) or on Bls12_377_Fq29 (13 x 29, R = 2^406: 337 instead of 378 multiply-adds per product).  The 29-bit shape has one bit
// less lazy headroom, which costs two carry passes in te_tail (and two more in the full addition); the birational map itself --
// init-time work -- always runs in the 14 x 28 shape and k_te_convert re-radixes its records (te_record_to_29).
//
// Extended points reuse XyzzT<Fe>: x = X, y = Y, zz = Z, zzz = T with X Y = Z T; all four are class M (strictly normalized
// limbs, value < 1.5p).  Base records hold (Y - X, Y + X, 2 d X Y), canonical: what the mixed addition multiplies by, so the
// hot loop neither forms them (42 limb operations per addition) nor selects between them for a negated base -- -(X, Y) =
// (-X, Y) swaps the first two, and a lane that reads its record out of LDS swaps two ADDRESSES instead of 28 registers.
#pragma once
#include "curve.hpp"

namespace msm {

#ifndef MSM_TE_LIMBS29
#define MSM_TE_LIMBS29 1   // 0 = the twisted-Edwards path on 14 x 28 limbs as in rounds 2-5 (A/B: profiles/r06_ab_limbs29.txt)
#endif
#if MSM_TE_LIMBS29
using TeFq = Bls12_377_Fq29;
#else
using TeFq = Bls12_377_Fq;
#endif

// the Edwards constant 2d in the representation of F
template <class F>
struct TeConst;
template <>
struct TeConst<Bls12_377_Fq> {
  static MSM_HD void k2d(Fe& r) { fe_set(r, Bls12_377_Te::K2D); }
};
template <>
struct TeConst<Bls12_377_Fq29> {
  static MSM_HD void k2d(Fe& r) { fe_set(r, Bls12_377_Cross::K2D29); }
};

struct TeAffine {
  Fe ymx, ypx, td;   // Y - X, Y + X, 2 d X Y
};
// Device record: ONE FIELD PER 64-B SECTOR (56 B + 8 B of padding), 192 B.  A gather touches exactly three sectors, and a
// kernel that receives the record sector by sector (k_accumulate_glds) applies the negation swap by choosing which sector it
// reads Y - X from.
#ifndef TE_REC_PAD256
#define TE_REC_PAD256 0   // A/B only (profiles/r03_ab_power.txt): 1 = records 256 bytes apart, so that none straddles a 256-byte boundary
#endif
struct alignas(TE_REC_PAD256 ? 256 : 64) TeAffineDev {
  Fe ymx;
  uint32_t pad0[2];
  Fe ypx;
  uint32_t pad1[2];
  Fe td;
  uint32_t pad2[2];
  MSM_HD TeAffine get() const {
    TeAffine t;
    t.ymx = ymx;
    t.ypx = ypx;
    t.td = td;
    return t;
  }
  MSM_HD void set(const TeAffine& t) {
    ymx = t.ymx;
    ypx = t.ypx;
    td = t.td;
    pad0[0] = pad0[1] = pad1[0] = pad1[1] = pad2[0] = pad2[1] = 0;
  }
};
static_assert(sizeof(TeAffineDev) == (TE_REC_PAD256 ? 256 : 192), "device twisted-Edwards base layout");

template <class F>
MSM_HD void te_set_identity(Xyzz& r) {
  fe_zero(r.x);
  fe_set(r.y, F::ONE);
  fe_set(r.zz, F::ONE);
  fe_zero(r.zzz);
}

// An addition whose denominator vanished leaves Z = 0 (valid points never have Z = 0).
template <class F>
MSM_HD bool te_failed(const Xyzz& a) {
  return fe_is_zero_M<F>(a.zz);
}

// Shared tail: from A, B, C (class M) and D (limbs < 2^(B+1), value < 3p) produce the sum.
//   E = B - A, F = D - C, G = D + C, H = B + A;  X3 = E F, Y3 = G H, T3 = E H, Z3 = F G.
// Limb bounds in units of 2^B (class M = 1):  E < 3, F < 4, G < 3, H < 2.  14 x 28 multiplies limbs < 4 x 4; 13 x 29 needs
// limb_a * limb_b < 3.9, so F and H are carried first (-> 1 + eps): E F' , G H', E H' < 3 and F' G < 3.  The top limbs are not
// shortened by a carry pass (F: 4p >> 348 = 3.4 * 2^29) and enter two terms of a column; tools/limb_bounds29.py has the exact sums.
template <class F>
MSM_HD void te_tail(Xyzz& r, const Fe& A, const Fe& B, const Fe& C, const Fe& D, const Modulus<F>& md) {
  constexpr int N = F::N;
  Fe e, f, g, h;
  fe_sub<N>(e, B, A, F::BIAS2_L);   // (0.5p, 3.5p), limbs < 2^B + 2^(B+1)
  fe_sub<N>(f, D, C, F::BIAS2_L);   // (0.5p, 5p),   limbs < 2^(B+1) + 2^(B+1)
  fe_add<N>(g, D, C);                // < 4.5p,       limbs < 2^(B+1) + 2^B
  fe_add<N>(h, B, A);                // < 3p,         limbs < 2^(B+1)
  if constexpr (F::B == 29) {
    fe_carry<N, F::B>(f);
    fe_carry<N, F::B>(h);
  }
  fe_mul<F>(r.x, e, f, md);       // 3.5p * 5p
  fe_mul<F>(r.y, g, h, md);
  fe_mul<F>(r.zzz, e, h, md);
  fe_mul<F>(r.zz, f, g, md);      // 5p * 4.5p < 2^10 p^2
}

// acc += (+/-) base   (7M).  -(X, Y) = (-X, Y): Y - X and Y + X trade places and 2dXY changes sign.
// SWAPPED = the caller has already exchanged ymx / ypx of a negated base (k_accumulate_glds does it with the LDS read
// addresses); only the sign of 2dXY is left to apply.
template <class F, bool SWAPPED = false>
MSM_HD void te_madd(Xyzz& acc, const TeAffine& b, bool negate, const Modulus<F>& md) {
  constexpr int N = F::N;
  const LaneMask neg = lane_mask(negate);
  Fe ymx = b.ymx, ypx = b.ypx, td, ntd;
  if (!SWAPPED) {
    fe_cmov<N>(ymx, b.ypx, neg);
    fe_cmov<N>(ypx, b.ymx, neg);
  }
  fe_neg<N>(ntd, b.td, F::BIAS2_L);       // (p, 2p], limbs < 2^(B+1)
  td = b.td;
  fe_cmov<N>(td, ntd, neg);
  Fe a1, b1, A, B, C, D;
  fe_sub<N>(a1, acc.y, acc.x, F::BIAS2_L);   // (0, 4p), limbs < 2^B + 2^(B+1)
  fe_add<N>(b1, acc.y, acc.x);                // < 4p,    limbs < 2^(B+1)
  fe_mul<F>(A, a1, ymx, md);               // 4p * p     (13 x 29: limbs 3 x 1)
  fe_mul<F>(B, b1, ypx, md);               //            (2 x 1)
  fe_mul<F>(C, acc.zzz, td, md);           //            (1 x 2)
  fe_dbl<N>(D, acc.zz);                       // < 3p, limbs < 2^(B+1)
  te_tail<F>(acc, A, B, C, D, md);
}

// acc += b, both extended (9M: add-2008-hwcd-3 with k = 2d).  Unified: b may equal acc.
template <class F>
MSM_HD void te_add(Xyzz& acc, const Xyzz& b, const Modulus<F>& md) {
  constexpr int N = F::N;
  Fe a1, a2, b1, b2, A, B, C, D, kt, zz, k;
  fe_sub<N>(a1, acc.y, acc.x, F::BIAS2_L);   // (0, 4p)
  fe_sub<N>(a2, b.y, b.x, F::BIAS2_L);
  fe_add<N>(b1, acc.y, acc.x);                // < 4p
  fe_add<N>(b2, b.y, b.x);
  if constexpr (F::B == 29) {   // limbs 3 x 3 and 2 x 2 would overflow a column of 13: one operand of each product is carried
    fe_carry<N, F::B>(a1);
    fe_carry<N, F::B>(b1);
  }
  TeConst<F>::k2d(k);
  fe_mul<F>(kt, b.zzz, k, md);
  fe_mul<F>(zz, acc.zz, b.zz, md);
  fe_mul<F>(A, a1, a2, md);                // 4p * 4p
  fe_mul<F>(B, b1, b2, md);
  fe_mul<F>(C, acc.zzz, kt, md);
  fe_dbl<N>(D, zz);
  te_tail<F>(acc, A, B, C, D, md);
}

#if defined(__HIPCC__)
// ---- one addition by the four lanes of a quad ------------------------------------------------------------------------
// Where an MSM is a chain of DEPENDENT additions on a nearly idle chip (the fragment merge and the scan reduction of small and
// medium inputs: ~14 us per step for a lone wave, 25-40 steps), the 9 multiplications of the unified addition are spread over
// a quad: lane q holds coordinate q of both operands (0 X, 1 Y, 2 Z, 3 T -- the order they lie in memory) and of the result.
//   step 1   lane 0: A = (Y1 - X1)(Y2 - X2)   lane 1: B = (Y1 + X1)(Y2 + X2)   lane 2: Z1 Z2   lane 3: k T2
//   step 2   lane 3: C = T1 (k T2)            (the other lanes idle through it)
//   step 3   lane 0: X3 = E F   lane 1: Y3 = G H   lane 2: Z3 = F G   lane 3: T3 = E H
// Three multiplications deep instead of nine; operands travel by DPP quad permutes (X <-> Y between lanes 0 and 1 before
// step 1, A, B, Z1 Z2, C to every lane before step 3).  Same bounds as te_add / te_tail.  Returns the lane's coordinate of
// a + b; all four lanes of the quad must call it together.
template <class F>
__device__ __forceinline__ void te_add_quad(Fe& a, const Fe& b, uint32_t q, const Modulus<F>& md) {
  constexpr int N = F::N;
  Fe pa, pb, u, v, t0, t1;
  fe_quad_perm<0xB1, N>(pa, a);                  // lanes 0 <-> 1, 2 <-> 3
  fe_quad_perm<0xB1, N>(pb, b);
  // lane 0 (own X, partner Y): Y - X;  lane 1 (own Y, partner X): Y + X
  fe_sub<N>(t0, pa, a, F::BIAS2_L);             // (0, 4p), limbs < 2^B + 2^(B+1)
  fe_add<N>(t1, a, pa);                          // < 4p, limbs < 2^(B+1)
  fe_select<N>(u, t1, t0, q == 0);
  fe_sub<N>(t0, pb, b, F::BIAS2_L);
  fe_add<N>(t1, b, pb);
  fe_select<N>(v, t1, t0, q == 0);
  if constexpr (F::B == 29) fe_carry<N, F::B>(u);   // as te_add: one operand of the 3 x 3 / 2 x 2 products
  // lane 2: Z1, Z2;  lane 3: k, T2
  Fe k;
  TeConst<F>::k2d(k);
  fe_select<N>(t0, a, k, q == 3);
  fe_select<N>(u, u, t0, q >= 2);
  fe_select<N>(v, v, b, q >= 2);
  Fe r1, r2;
  fe_mul<F>(r1, u, v, md);                    // A | B | Z1 Z2 | k T2
  fe_mul<F>(r2, a, r1, md);                   // lane 3: C = T1 (k T2)
  fe_select<N>(r1, r1, r2, q == 3);
  Fe A, B, C, D, Z;
  fe_quad_perm<0x00, N>(A, r1);
  fe_quad_perm<0x55, N>(B, r1);
  fe_quad_perm<0xAA, N>(Z, r1);
  fe_quad_perm<0xFF, N>(C, r1);
  fe_dbl<N>(D, Z);                               // < 3p, limbs < 2^(B+1)
  Fe e, f, g, h;
  fe_sub<N>(e, B, A, F::BIAS2_L);               // as te_tail
  fe_sub<N>(f, D, C, F::BIAS2_L);
  fe_add<N>(g, D, C);
  fe_add<N>(h, B, A);
  if constexpr (F::B == 29) {
    fe_carry<N, F::B>(f);
    fe_carry<N, F::B>(h);
  }
  // lane 0: E F   lane 1: G H   lane 2: F G   lane 3: E H
  fe_select<N>(u, e, g, q == 1);
  fe_select<N>(u, u, f, q == 2);
  fe_select<N>(v, h, f, q == 0);
  fe_select<N>(v, v, g, q == 2);
  fe_mul<F>(a, u, v, md);
}
#endif

template <class F>
MSM_HD void te_dbl(Xyzz& acc, const Modulus<F>& md) {
  const Xyzz b = acc;
  te_add<F>(acc, b, md);
}

// ---- the birational map ---------------------------------------------------------------------------------------
// SW affine (canonical coordinates) -> what the image needs:  u = S (x + 1), v = S y, w = u + 1, den = v w.
// den == 0 (mod p) <=> the point has no image (y = 0: the three 2-torsion points; u = -1: two more points).
template <class F>
MSM_HD void te_map_prepare(Fe& u, Fe& v, Fe& w, Fe& den, const Affine& p, const Modulus<F>& md) {
  Fe s, one, xp1;
  fe_set(s, Bls12_377_Te::S);
  fe_set(one, F::ONE);
  fe_add(xp1, p.x, one);          // < 2.5p, limbs < 2^29
  fe_mul<F>(u, s, xp1, md);
  fe_mul<F>(v, s, p.y, md);
  fe_add(w, u, one);              // < 3p, limbs < 2^29
  fe_mul<F>(den, v, w, md);
}

// With inv = 1/den:  X = FSC u w inv,  Y = (u - 1) v inv;  the record is (Y - X, Y + X, 2d X Y), all canonical.
template <class F>
MSM_HD void te_map_finish(TeAffine& out, const Fe& u, const Fe& v, const Fe& w, const Fe& inv, const Modulus<F>& md) {
  Fe t, c, one, um1, X, Y;
  fe_set(one, F::ONE);
  fe_mul<F>(t, u, w, md);
  fe_mul<F>(t, t, inv, md);
  fe_set(c, Bls12_377_Te::FSC);
  fe_mul<F>(X, t, c, md);
  fe_sub(um1, u, one, F::BIAS2_28);   // (0.5p, 3.5p), limbs < 2^28 + 2^29
  fe_mul<F>(t, um1, v, md);
  fe_mul<F>(Y, t, inv, md);
  fe_mul<F>(t, X, Y, md);
  fe_set(c, Bls12_377_Te::K2D);
  fe_mul<F>(out.td, t, c, md);
  fe_sub(out.ymx, Y, X, F::BIAS2_28);  // class M inputs: (0.5p, 3.5p), limbs < 2^28 + 2^29
  fe_add(out.ypx, Y, X);               // < 3p, limbs < 2^29
  fe_reduce<F>(out.ymx);
  fe_reduce<F>(out.ypx);
  fe_reduce<F>(out.td);
}

// Extended twisted-Edwards point -> short-Weierstrass XYZZ (affine, ZZ = ZZZ = 1).  One inversion: host side / rare.
//   u = (Z + Y)/(Z - Y),  v = FSC u Z / X;   x = sqrt3 u - 1,  y = sqrt3 v.
// (0, 1) is the identity -> infinity; (0, -1) is the image of the 2-torsion point (-1, 0).  The caller has checked Z != 0.
template <class F>
MSM_HD void te_to_sw(Xyzz& out, const Xyzz& a, const Modulus<F>& md) {
  using E = FpEl<F>;
  Fe one;
  fe_set(one, F::ONE);
  if (fe_is_zero_M<F>(a.x)) {
    Fe d;
    fe_sub(d, a.y, a.zz, F::BIAS2_28);
    if (fe_is_zero_slow<F>(d)) {
      xyzz_set_inf<E>(out);
    } else {
      fe_neg(out.x, one, F::BIAS2_28);
      fe_reduce<F>(out.x);
      fe_zero(out.y);
      out.zz = one;
      out.zzz = one;
    }
    return;
  }
  Fe n, dn, den, inv, t, c;
  fe_add(n, a.zz, a.y);                     // < 3p, limbs < 2^29
  fe_sub(dn, a.zz, a.y, F::BIAS2_28);       // (0.5p, 3.5p)
  fe_mul<F>(den, dn, a.x, md);
  fe_inv<F>(inv, den, md);
  fe_mul<F>(t, n, a.x, md);
  fe_mul<F>(t, t, inv, md);                 // u
  fe_set(c, Bls12_377_Te::SQRT3);
  fe_mul<F>(t, t, c, md);
  fe_sub(out.x, t, one, F::BIAS2_28);       // (0.5p, 3.5p)
  fe_carry(out.x);
  fe_mul<F>(t, n, a.zz, md);
  fe_mul<F>(t, t, inv, md);                 // u Z / X
  fe_set(c, Bls12_377_Te::FSC_SQRT3);
  fe_mul<F>(out.y, t, c, md);
  out.zz = one;
  out.zzz = one;
}

// result = sum_w 2^(c*w) * sums[w] on the twisted-Edwards image, mapped back to short-Weierstrass XYZZ (generic arithmetic;
// the engine's host tail uses the 64-bit twin in host_fold64.hpp, tests/test_fold64_host.py compares the two).
// false: an addition hit a vanishing denominator (possible only off the odd-order subgroup).
template <class F>
MSM_HD bool fold_windows_te(Xyzz& out, const Xyzz* sums, int windows, int c, const Modulus<F>& md) {
  Xyzz acc;
  te_set_identity<F>(acc);
  for (int w = windows - 1; w >= 0; w--) {
    if (w != windows - 1)
      for (int i = 0; i < c; i++) {
        te_dbl<F>(acc, md);
        if (te_failed<F>(acc)) return false;
      }
    if (te_failed<F>(sums[w])) return false;
    te_add<F>(acc, sums[w], md);
    if (te_failed<F>(acc)) return false;
  }
  te_to_sw<F>(out, acc, md);
  return true;
}

// ---- between the limb shapes ---------------------------------------------------------------------------------------------
// A canonical 14 x 28 record (what te_map_finish leaves) in the representation of F; the identity for F = Bls12_377_Fq.
template <class F>
MSM_HD void te_record_to(TeAffine& r, const TeAffine& a, const Modulus<F>& md) {
  if constexpr (F::B == 29) {
    fe_28_to_29(r.ymx, a.ymx, md);
    fe_28_to_29(r.ypx, a.ypx, md);
    fe_28_to_29(r.td, a.td, md);
    // canonical, like the 14 x 28 records: equal points stay bit-identical and the bounds of te_madd start from < p
    fe_reduce<F>(r.ymx);
    fe_reduce<F>(r.ypx);
    fe_reduce<F>(r.td);
  } else {
    r = a;
  }
}
// An extended point held in the representation of F as a 14 x 28 point (class M): what te_to_sw / fold_windows_te and the host take.
template <class F>
MSM_HD void te_point_to_28(Xyzz& r, const Xyzz& a, const Modulus<Bls12_377_Fq>& md28) {
  if constexpr (F::B == 29) {
    fe_29_to_28(r.x, a.x, md28);
    fe_29_to_28(r.y, a.y, md28);
    fe_29_to_28(r.zz, a.zz, md28);
    fe_29_to_28(r.zzz, a.zzz, md28);
  } else {
    r = a;
  }
}
template <class F>
MSM_HD void te_point_from_28(Xyzz& r, const Xyzz& a, const Modulus<F>& md) {
  if constexpr (F::B == 29) {
    Xyzz t = a;
    fe_reduce<Bls12_377_Fq>(t.x);
    fe_reduce<Bls12_377_Fq>(t.y);
    fe_reduce<Bls12_377_Fq>(t.zz);
    fe_reduce<Bls12_377_Fq>(t.zzz);
    fe_28_to_29(r.x, t.x, md);
    fe_28_to_29(r.y, t.y, md);
    fe_28_to_29(r.zz, t.zz, md);
    fe_28_to_29(r.zzz, t.zzz, md);
  } else {
    r = a;
  }
}

}  // namespace msm
