// fp2pair.hpp -- Fp2 arithmetic with ONE ELEMENT SPREAD OVER TWO NEIGHBOURING LANES (device only).
//
// Why: the G2 walking kernels with a whole Fp2 point per lane (Fp2El, curve.hpp) need 356 registers for one mixed addition, so
// they run ONE wave per SIMD, and a lone wave issues one VALU instruction per ~5.3 cycles where two or more waves issue one per
// ~4.1 (profiles/r03_pmc_k_accumulate_g2.json: 77 % VALU busy at 2.35 GHz -- issue-limited, not power-limited).  Here lane 2k of a
// wave holds c0 and lane 2k + 1 holds c1 of every Fp2 value of "pair" k: a coordinate is 14 VGPRs instead of 28, the group law of
// curve.hpp -- which is generic over the coordinate policy -- runs unchanged with T = Fe, and the kernels fit two waves per SIMD.
//
// The product (ARK ff/src/fields/models/quadratic_extension.rs:641-652 computes the same two components by Karatsuba):
//     (a0 + a1 u)(b0 + b1 u) = (a0 b0 + BETA a1 b1) + (a1 b0 + a0 b1) u,        BETA = -NEG_BETA
// is ONE fused dual product per lane (fe_mul2: two limb products, one Montgomery reduction),
//     lane h:   r_h = a_h * b0 + A * Z,      even: A = NEG_BETA a1, Z = K p - b1        odd: A = a0, Z = b1
// with three DPP quad permutes per limb fetching b0 (quad_perm [0,0,2,2]), b1 ([1,1,3,3]) and the partner's a ([1,0,3,2]).  The
// operands of each fe_mul2 are EXACTLY those of Fp2El::mul_c / sqr_c / mul (same biases, same scaling), so every limb the paired
// form produces equals the one-lane form's: bounds are those checked on the host for Fp2El (tests/test_field_host.py), and
// tests/test_gpu_devtest.py compares the two limb for limb on the device.
// Cost: 84 (BETA = -5) / 70 (BETA = -1) cheap instructions per product on top of the 588 multiply-adds, and a squaring costs a
// product (the one-lane form saves a third of the u-part) -- paid for by the second wave.
#pragma once
#include "curve.hpp"

#if defined(__HIPCC__)
namespace msm {

template <class F, int NEG_BETA>
struct PairMd : Modulus<F> {
  uint32_t half;   // 0: this lane holds c0, 1: c1
  uint32_t kfac;   // what the partner's a is scaled by: NEG_BETA on the even lane (BETA a1, sign on the b side), 1 on the odd lane
  LaneMask odd;
  __device__ __forceinline__ PairMd() : Modulus<F>() {
    half = __lane_id() & 1u;
    kfac = half ? 1u : (uint32_t)NEG_BETA;
    odd = lane_mask(half != 0);
  }
};

template <int CTRL>
__device__ __forceinline__ void fe_dpp(Fe& r, const Fe& a) {
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.v[i], CTRL, 0xf, 0xf, true);
}
constexpr int DPP_PAIR_LO = 0xA0;     // quad_perm [0,0,2,2]: the even lane's value to both lanes of a pair
constexpr int DPP_PAIR_HI = 0xF5;     // quad_perm [1,1,3,3]: the odd lane's value
constexpr int DPP_PAIR_SWAP = 0xB1;   // quad_perm [1,0,3,2]: the partner's value

template <class F, int NEG_BETA>
struct Fp2PairEl {
  using Fld = F;
  using T = Fe;       // what a lane holds: its half of the element
  using Md = PairMd<F, NEG_BETA>;
  using Whole = Fp2El<F, NEG_BETA>;

  // r = a * b for carried operands (limbs < 2^28 + 16): Fp2El::mul_c.  B_BIG: b of value <= 18p (bias 32p), else class M (bias 2p).
  template <bool B_BIG>
  static __device__ __forceinline__ void mul_c(T& r, const T& a, const T& b, const Md& md) {
    Fe y, z, nz, pa;
    fe_dpp<DPP_PAIR_LO>(y, b);
    fe_dpp<DPP_PAIR_HI>(z, b);
    if (B_BIG) fe_neg(nz, z, F::BIAS32_29); else fe_neg(nz, z, F::BIAS2_28);
    fe_cmov(nz, z, md.odd);                                   // even: K p - b1, odd: b1
    fe_dpp<DPP_PAIR_SWAP>(pa, a);
    if (NEG_BETA != 1) {
#pragma unroll
      for (int i = 0; i < NL; i++) pa.v[i] *= md.kfac;         // even: NEG_BETA a1 (limbs < 5 * (2^28 + 16)), odd: a0
    }
    fe_mul2<F>(r, a, y, pa, nz, md);
  }
  // (a0 + a1 u)^2: the same product with b = a (Fp2El::sqr_c forms the u part as a0 * 2 a1 -- the same column sums)
  static __device__ __forceinline__ void sqr_c(T& r, const T& a, const Md& md) { mul_c<true>(r, a, a, md); }
  static __device__ __forceinline__ void prep(T& a) { fe_carry(a); }
  // r = a b - c d as a stored coordinate (Fp2El::mul_sub_c)
  static __device__ __forceinline__ void mul_sub_c(T& r, const T& a, const T& b, const T& c, const T& d, const Md& md) {
    T t1, t2;
    mul_c<true>(t1, a, b, md);
    mul_c<false>(t2, c, d, md);
    fe_sub(r, t1, t2, F::BIAS2_28);
    fe_carry(r);
  }
  // any lazy operands (Fp2El::mul): rare paths only (doubling, same-x)
  static __device__ __forceinline__ void mul(T& r, const T& a, const T& b, const Md& md) {
    Fe ac = a, bc = b, bw = b, s, t, y, z, pa;
    fe_carry(ac);
    fe_carry(bc);
    fe_weak_reduce<F>(bw);                                    // (meaningful on the odd lane: b1 < 3p, normalized)
#pragma unroll
    for (int i = 0; i < NL; i++) s.v[i] = bw.v[i] * (uint32_t)NEG_BETA;
    fe_neg(t, s, F::BIAS16_31);                               // BETA b1 as (p, 16p]
    fe_carry(t);
    fe_dpp<DPP_PAIR_LO>(y, bc);                               // b0, carried
    fe_dpp<DPP_PAIR_HI>(z, t);                                // BETA b1 (the odd lane's t)
    fe_cmov(z, bw, md.odd);                                   // odd: b1, weakly reduced
    fe_dpp<DPP_PAIR_SWAP>(pa, ac);
    fe_mul2<F>(r, ac, y, pa, z, md);
  }
  static __device__ __forceinline__ void sqr(T& r, const T& a, const Md& md) { mul(r, a, a, md); }
  static __device__ __forceinline__ void mul2(T& r, const T& a, const T& b, const T& c, const T& d, const Md& md) {
    T t1, t2;
    mul(t1, a, b, md);
    mul(t2, c, d, md);
    fe_add(r, t1, t2);
    fe_carry(r);
  }
  static __device__ __forceinline__ void add(T& r, const T& a, const T& b) { fe_add(r, a, b); }
  static __device__ __forceinline__ void dbl(T& r, const T& a) { fe_dbl(r, a); }
  static __device__ __forceinline__ void sub(T& r, const T& a, const T& b, const uint32_t (&bias)[NL]) { fe_sub(r, a, b, bias); }
  static __device__ __forceinline__ void neg(T& r, const T& b, const uint32_t (&bias)[NL]) { fe_neg(r, b, bias); }
  static __device__ __forceinline__ void carry(T& r) { fe_carry(r); }
  // both halves zero: pair-uniform
  static __device__ __forceinline__ bool is_zero_M(const T& a) {
    const int z = fe_is_zero_M<F>(a) ? 1 : 0;
    return (z & __builtin_amdgcn_update_dpp(0, z, DPP_PAIR_SWAP, 0xf, 0xf, true)) != 0;
  }
  static __device__ __forceinline__ void cmov(T& r, const T& a, bool take) { fe_cmov(r, a, take); }
  static __device__ __forceinline__ void set_one(T& r) {
    const uint32_t odd = __lane_id() & 1u;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = odd ? 0u : F::ONE[i];
  }
  static __device__ __forceinline__ void zero(T& r) { fe_zero(r); }
};

// ---- a lane's half of the records in memory (layouts of msm_types.hpp: every Fp2 coordinate is c0 | c1, 56 bytes each) ----------
// half h of coordinate q of an XYZZ record / of an affine record
template <class P>
__device__ __forceinline__ const Fe* pair_coord(const P* rec, uint32_t q, uint32_t h) { return reinterpret_cast<const Fe*>(rec) + 2 * q + h; }
template <class P>
__device__ __forceinline__ Fe* pair_coord(P* rec, uint32_t q, uint32_t h) { return reinterpret_cast<Fe*>(rec) + 2 * q + h; }

// 56 bytes that are 8-byte aligned: seven 8-byte accesses
__device__ __forceinline__ Fe fe_load8(const Fe* p) {
  Fe r;
  const uint2* s = reinterpret_cast<const uint2*>(p);
#pragma unroll
  for (int k = 0; k < NL / 2; k++) {
    const uint2 v = s[k];
    r.v[2 * k] = v.x;
    r.v[2 * k + 1] = v.y;
  }
  return r;
}
__device__ __forceinline__ void fe_store8(Fe* p, const Fe& a) {
  uint2* d = reinterpret_cast<uint2*>(p);
#pragma unroll
  for (int k = 0; k < NL / 2; k++) d[k] = make_uint2(a.v[2 * k], a.v[2 * k + 1]);
}

}  // namespace msm
#endif
