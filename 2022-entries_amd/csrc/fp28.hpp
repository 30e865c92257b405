// fp28.hpp -- 384-bit prime-field arithmetic for gfx950, unsaturated radix-2^28 limbs.
//
// Why this shape (measured on MI355X, tools/ubench_valu.hip, profiles/r01_ubench_valu.txt):
// v_mad_u64_u32 issues at the same ~4.3 cycles/wave as v_addc_co_u32, and gfx950 needs two wait
// states between a VALU that writes a carry (VCC/SGPR) and the VALU that consumes it.  A saturated
// 12x32-bit Montgomery product therefore costs a carry instruction (plus hazard padding) per
// multiply-add.  With 14 limbs of 28 bits every partial product is < 2^60, a whole product-scanning
// column (<= 14 a*b terms + 14 m*p terms) fits a 64-bit accumulator, and the multiplication is one
// dependent chain of v_mad_u64_u32 with NO carry instructions.  Additions and subtractions are
// limb-wise (no carry chain either); values are kept lazily reduced and only bounded, see below.
//
// The same header compiles for the host (plain C++), which is how tests/ check the limb-bound
// analysis (MSM_CHECK) and how the host-side window fold (host_fold.cpp) shares the arithmetic.
//
// Reference behaviour being re-implemented (not translated): sppark mont_t mul/add/sub
// (SPK ff/mont_t.cuh:385-425, 187-212, 278-299) and Matter Labs' lazily reduced field
// (ML ff_dispatch_st.cuh:234-481).  The ABI Montgomery radix (2^384) is converted at the
// boundary with CIN/COUT.
//
// Representation.  value(a) = sum a.v[i] * 2^(28 i), Montgomery form x*R mod p with R = 2^392.
//   "normalized": v[0..12] < 2^28 (v[13] holds what is left).
//   mul inputs: every limb < 2^30 and value(a)*value(b) <= 2^10 p^2 (e.g. both values < 32p).
//   class M ("mul output"): normalized, value < p + a*b/R < 1.5p  (2^10 p^2 / 2^392 < p/2 for p < 2^381).
//
// A second limb shape, per FIELD (the constants struct F carries N = limbs in use, B = bits per limb, NRED = Montgomery steps):
// BLS12-377's 377 bits are 13 x 29, and Bls12_377_Fq29 runs fe_mul on 13 limbs with 14 reduction steps (R = 2^406) -- 169 + 168 = 337
// multiply-adds instead of 196 + 182 = 378.  The 14th step is not optional: with R = 2^377, p/R = 0.84 and the product of two lazily
// reduced operands (3p x 4p) comes out at 11p, not < 2p.  A column of 13 a*b + 13 m*p terms fits 64 bits only while
// 13 * limb_a * limb_b < 51 * 2^58, i.e. one bit less slack than 14 x 28 has: the twisted-Edwards law (te.hpp) pays for it with two
// carry passes per addition (tools/limb_bounds29.py proves the column bounds, MSM_CHECK builds check them on every product).
// Fe keeps 14 words either way (word 13 of a 13-limb value is 0), so records in memory and every kernel are shared.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MSM_HD __host__ __device__ __forceinline__
#else
#define MSM_HD inline
#endif

#ifndef MSM_CHECK
#define MSM_CHECK(cond) ((void)0)
#define MSM_CHECK_COL_BEGIN() ((void)0)
#define MSM_CHECK_COL_ADD(x) ((void)0)
#define MSM_CHECK_COL_END(col) ((void)0)
#else
// host-side bound checker: re-accumulate every column in 128 bits and require that it fits 64
#define MSM_CHECK_COL_BEGIN() unsigned __int128 chk_col_ = col
#define MSM_CHECK_COL_ADD(x) chk_col_ += (x)
#define MSM_CHECK_COL_END(col) MSM_CHECK(chk_col_ == (unsigned __int128)(col))
#endif

namespace msm {

constexpr int NL = 14;
constexpr int LB = 28;
constexpr uint32_t LMASK = 0x0fffffffu;

#include "field_consts.inc"

struct Fe {
  uint32_t v[NL];
};

// Keep a modulus limb in a scalar register and opaque to the optimiser, so that m*p[j] stays a
// single v_mad_u64_u32 with an SGPR operand instead of being strength-reduced into shift/add chains.
MSM_HD uint32_t opaque_u32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+s"(x));
#endif
  return x;
}

template <class F>
struct Modulus {
  uint32_t p[NL];
  MSM_HD Modulus() {
#pragma unroll
    for (int i = 0; i < NL; i++) p[i] = opaque_u32(F::P[i]);
  }
};

MSM_HD void fe_set(Fe& r, const uint32_t (&c)[NL]) {
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = c[i];
}

MSM_HD void fe_zero(Fe& r) {
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = 0;
}

// col += sum_{i<n} x[i]*y[i] as ONE uninterrupted chain of v_mad_u64_u32 (n <= 14, a compile-time constant after
// unrolling).  Written as inline assembly on the device because hipcc otherwise splits such a chain every few terms
// with a v_lshl_add_u64/v_mov pair (it reassociates the 64-bit sum): 34 extra VALU instructions per multiplication,
// 4-5 % of its issue time (tools/ubench_mul.hip).  YS = true: the y operands are wave-uniform (modulus limbs in SGPRs).
#if defined(__HIP_DEVICE_COMPILE__)
#define MSM_MAD(x, y) "v_mad_u64_u32 %0, vcc, %" #x ", %" #y ", %0\n\t"
#define MSM_MADS_1 MSM_MAD(1, 2)
#define MSM_MADS_2 MSM_MADS_1 MSM_MAD(3, 4)
#define MSM_MADS_3 MSM_MADS_2 MSM_MAD(5, 6)
#define MSM_MADS_4 MSM_MADS_3 MSM_MAD(7, 8)
#define MSM_MADS_5 MSM_MADS_4 MSM_MAD(9, 10)
#define MSM_MADS_6 MSM_MADS_5 MSM_MAD(11, 12)
#define MSM_MADS_7 MSM_MADS_6 MSM_MAD(13, 14)
#define MSM_MADS_8 MSM_MADS_7 MSM_MAD(15, 16)
#define MSM_MADS_9 MSM_MADS_8 MSM_MAD(17, 18)
#define MSM_MADS_10 MSM_MADS_9 MSM_MAD(19, 20)
#define MSM_MADS_11 MSM_MADS_10 MSM_MAD(21, 22)
#define MSM_MADS_12 MSM_MADS_11 MSM_MAD(23, 24)
#define MSM_MADS_13 MSM_MADS_12 MSM_MAD(25, 26)
#define MSM_MADS_14 MSM_MADS_13 MSM_MAD(27, 28)
#define MSM_OPS_1(YC) "v"(x[0]), YC(y[0])
#define MSM_OPS_2(YC) MSM_OPS_1(YC), "v"(x[1]), YC(y[1])
#define MSM_OPS_3(YC) MSM_OPS_2(YC), "v"(x[2]), YC(y[2])
#define MSM_OPS_4(YC) MSM_OPS_3(YC), "v"(x[3]), YC(y[3])
#define MSM_OPS_5(YC) MSM_OPS_4(YC), "v"(x[4]), YC(y[4])
#define MSM_OPS_6(YC) MSM_OPS_5(YC), "v"(x[5]), YC(y[5])
#define MSM_OPS_7(YC) MSM_OPS_6(YC), "v"(x[6]), YC(y[6])
#define MSM_OPS_8(YC) MSM_OPS_7(YC), "v"(x[7]), YC(y[7])
#define MSM_OPS_9(YC) MSM_OPS_8(YC), "v"(x[8]), YC(y[8])
#define MSM_OPS_10(YC) MSM_OPS_9(YC), "v"(x[9]), YC(y[9])
#define MSM_OPS_11(YC) MSM_OPS_10(YC), "v"(x[10]), YC(y[10])
#define MSM_OPS_12(YC) MSM_OPS_11(YC), "v"(x[11]), YC(y[11])
#define MSM_OPS_13(YC) MSM_OPS_12(YC), "v"(x[12]), YC(y[12])
#define MSM_OPS_14(YC) MSM_OPS_13(YC), "v"(x[13]), YC(y[13])
#define MSM_CHAIN_CASE(n)                                                    \
  case n:                                                                    \
    if (YS)                                                                  \
      asm(MSM_MADS_##n : "+v"(col) : MSM_OPS_##n("s") : "vcc");              \
    else                                                                     \
      asm(MSM_MADS_##n : "+v"(col) : MSM_OPS_##n("v") : "vcc");              \
    break;
#endif

template <bool YS>
MSM_HD void mad_chain(uint64_t& col, const uint32_t (&x)[NL], const uint32_t (&y)[NL], int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  switch (n) {
    MSM_CHAIN_CASE(1)
    MSM_CHAIN_CASE(2)
    MSM_CHAIN_CASE(3)
    MSM_CHAIN_CASE(4)
    MSM_CHAIN_CASE(5)
    MSM_CHAIN_CASE(6)
    MSM_CHAIN_CASE(7)
    MSM_CHAIN_CASE(8)
    MSM_CHAIN_CASE(9)
    MSM_CHAIN_CASE(10)
    MSM_CHAIN_CASE(11)
    MSM_CHAIN_CASE(12)
    MSM_CHAIN_CASE(13)
    MSM_CHAIN_CASE(14)
    default:
      break;
  }
#else
  for (int i = 0; i < n; i++) col += (uint64_t)x[i] * y[i];
#endif
}

// ~x & MASK in one instruction (v_bfi_b32 D = (S0 & S1) | (~S0 & S2) with S1 = 0): hipcc would emit v_not + v_and
template <uint32_t MASK>
MSM_HD uint32_t not_and_mask(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t d;
  asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(d) : "v"(x), "s"(MASK));
  return d;
#else
  return ~x & MASK;
#endif
}
MSM_HD uint32_t not_and_lmask(uint32_t x) { return not_and_mask<LMASK>(x); }

// The Montgomery step of column k: col += m_k * p_0 (which clears the low limb), then shift the column down.
// When p = 1 (mod 2^28) -- BLS12-377, whose p - 1 is divisible by 2^46 -- m_k = -col mod 2^28 and p_0 = 1, so
// (col + m_k) >> 28 = (col + 2^28 - 1) >> 28: one 64-bit add instead of a multiply-add (126 fewer per mixed addition), and
// m_k itself falls out of that sum: with s = col + 2^28 - 1,  ~s = -col - 2^28  (mod 2^32), so m_k = ~s & (2^28 - 1) --
// three instructions per column (v_lshl_add_u64, v_bfi_b32, v_lshrrev_b64) instead of four.
// MK_FROM_COL: the caller has not computed m_k yet (the p_0 = 1 path derives it here; otherwise it is (col * M0) & LMASK).
#define MSM_MONT_STEP(F, col, mk, md)                                              \
  do {                                                                             \
    constexpr uint32_t FM_ = (1u << F::B) - 1;                                     \
    if (F::P[0] == 1) {                                                            \
      const uint64_t s_ = (col) + FM_;                                             \
      (mk) = not_and_mask<FM_>((uint32_t)s_);                                      \
      MSM_CHECK((mk) == ((0u - (uint32_t)(col)) & FM_));                           \
      MSM_CHECK_COL_ADD(mk);                                                       \
      MSM_CHECK_COL_END((col) + (mk));                                             \
      MSM_CHECK((((col) + (mk)) & FM_) == 0 && (((col) + (mk)) >> F::B) == (s_ >> F::B)); \
      (col) = s_ >> F::B;                                                          \
    } else {                                                                       \
      (mk) = ((uint32_t)(col) * F::M0) & FM_;                                      \
      (col) += (uint64_t)(mk) * (md).p[0];                                         \
      MSM_CHECK_COL_ADD((unsigned __int128)(mk) * (md).p[0]);                      \
      MSM_CHECK_COL_END(col);                                                      \
      MSM_CHECK(((uint32_t)(col) & FM_) == 0);                                     \
      (col) >>= F::B;                                                              \
    }                                                                              \
  } while (0)

// r = a*b*R^-1 (mod p), class M.  Product scanning: column k gathers every a_i*b_j and m_i*p_j with
// i+j = k in ONE 64-bit accumulator; m_k clears the low B bits and the column is shifted down.  N limbs, NRED >= N reduction
// steps (R = 2^(B*NRED)): columns 0 .. NRED-1 produce the m_k, columns NRED .. NRED+N-1 the result.
// Bound (14 x 28): limbs < 2^30  =>  14*(2^30)^2 + 14*(2^28)^2 + carry < 2^64.
// Bound (13 x 29): 13 * limb_a * limb_b + 13 * 2^58 + carry < 2^64, i.e. limb_a * limb_b < 3.9 * 2^58 -- the callers' business (te.hpp).
template <class F>
MSM_HD void fe_mul(Fe& r, const Fe& a, const Fe& b, const Modulus<F>& md) {
  constexpr int N = F::N, NR = F::NRED, B = F::B;
  constexpr uint32_t MASK = (1u << B) - 1;
  static_assert(N <= NR && NR <= NL && N * B >= F::BITS, "limb shape");
  uint32_t m[NL], xs[NL], ys[NL];
  Fe t;
  uint64_t col = 0;
  if (B == 28) {
#pragma unroll
    for (int i = 0; i < N; i++) {
      MSM_CHECK(a.v[i] < (1u << 30) && b.v[i] < (1u << 30));
    }
  }
#pragma unroll
  for (int k = 0; k < NR; k++) {
    MSM_CHECK_COL_BEGIN();
    const int lo = k > N - 1 ? k - (N - 1) : 0, hi = k < N - 1 ? k : N - 1;
    int n = 0;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      xs[n] = a.v[i];
      ys[n] = b.v[k - i];
      n++;
      MSM_CHECK_COL_ADD((unsigned __int128)a.v[i] * b.v[k - i]);
    }
    mad_chain<false>(col, xs, ys, n);
    n = 0;
#pragma unroll
    for (int i = lo; i < k; i++) {
      xs[n] = m[i];
      ys[n] = md.p[k - i];
      n++;
      MSM_CHECK_COL_ADD((unsigned __int128)m[i] * md.p[k - i]);
    }
    mad_chain<true>(col, xs, ys, n);
    MSM_MONT_STEP(F, col, m[k], md);
  }
#pragma unroll
  for (int k = NR; k < NR + N - 1; k++) {
    MSM_CHECK_COL_BEGIN();
    const int lo = k - (N - 1);
    int n = 0;
#pragma unroll
    for (int i = lo; i < N; i++) {
      xs[n] = a.v[i];
      ys[n] = b.v[k - i];
      n++;
      MSM_CHECK_COL_ADD((unsigned __int128)a.v[i] * b.v[k - i]);
    }
    mad_chain<false>(col, xs, ys, n);
    n = 0;
#pragma unroll
    for (int i = lo; i < NR; i++) {
      xs[n] = m[i];
      ys[n] = md.p[k - i];
      n++;
      MSM_CHECK_COL_ADD((unsigned __int128)m[i] * md.p[k - i]);
    }
    mad_chain<true>(col, xs, ys, n);
    MSM_CHECK_COL_END(col);
    t.v[k - NR] = (uint32_t)col & MASK;
    col >>= B;
  }
  MSM_CHECK(col < (1ull << (B == 28 ? 28 : 30)));
  t.v[N - 1] = (uint32_t)col;
#pragma unroll
  for (int i = N; i < NL; i++) t.v[i] = 0;
  r = t;
}

// r = (a*b + c*d)*R^-1 (mod p), class M: two products share ONE Montgomery reduction (the "reduce(ab) + reduce(cd) =
// reduce(ab + cd)" saving of ML ec.cuh:531-536).  Bound: every limb < 2^29  =>  28*(2^29)^2 + 14*(2^28)^2 + carry < 2^64;
// values: a*b + c*d <= 2^10 p^2 as for fe_mul.
template <class F>
MSM_HD void fe_mul2(Fe& r, const Fe& a, const Fe& b, const Fe& c, const Fe& d, const Modulus<F>& md) {
  static_assert(F::N == NL && F::NRED == NL && F::B == LB, "fe_mul2 is written for the 14 x 28 shape");
  uint32_t m[NL], xs[NL], ys[NL];
  Fe t;
  uint64_t col = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    // (or: c up to 5 * (2^28 + 16) against d < 2^29 + 2^28 when a, b are carried -- Fp2El::mul_c; the column checks below
    //  hold the exact sums against 2^64 either way)
    MSM_CHECK((a.v[i] < (1u << 29) && b.v[i] < (1u << 29) && c.v[i] < (1u << 29) && d.v[i] < (1u << 29)) ||
              (a.v[i] < (1u << 28) + 16 && b.v[i] < (1u << 28) + 16 && c.v[i] < 5 * ((1u << 28) + 16) && d.v[i] < (3u << 28)));
  }
  // (two separately unrolled halves: one 27-trip loop is only partially unrolled by hipcc and then indexes
  //  registers dynamically)
#pragma unroll
  for (int k = 0; k < NL; k++) {
    MSM_CHECK_COL_BEGIN();
#pragma unroll
    for (int i = 0; i <= k; i++) {
      xs[i] = a.v[i];
      ys[i] = b.v[k - i];
      MSM_CHECK_COL_ADD((unsigned __int128)a.v[i] * b.v[k - i] + (unsigned __int128)c.v[i] * d.v[k - i]);
    }
    mad_chain<false>(col, xs, ys, k + 1);
#pragma unroll
    for (int i = 0; i <= k; i++) {
      xs[i] = c.v[i];
      ys[i] = d.v[k - i];
    }
    mad_chain<false>(col, xs, ys, k + 1);
#pragma unroll
    for (int i = 0; i < k; i++) {
      xs[i] = m[i];
      ys[i] = md.p[k - i];
      MSM_CHECK_COL_ADD((unsigned __int128)m[i] * md.p[k - i]);
    }
    mad_chain<true>(col, xs, ys, k);
    MSM_MONT_STEP(F, col, m[k], md);
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
    MSM_CHECK_COL_BEGIN();
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) {
      xs[i - (k - NL + 1)] = a.v[i];
      ys[i - (k - NL + 1)] = b.v[k - i];
      MSM_CHECK_COL_ADD((unsigned __int128)a.v[i] * b.v[k - i] + (unsigned __int128)c.v[i] * d.v[k - i]);
    }
    mad_chain<false>(col, xs, ys, 2 * NL - 1 - k);
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) {
      xs[i - (k - NL + 1)] = c.v[i];
      ys[i - (k - NL + 1)] = d.v[k - i];
    }
    mad_chain<false>(col, xs, ys, 2 * NL - 1 - k);
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) {
      xs[i - (k - NL + 1)] = m[i];
      ys[i - (k - NL + 1)] = md.p[k - i];
      MSM_CHECK_COL_ADD((unsigned __int128)m[i] * md.p[k - i]);
    }
    mad_chain<true>(col, xs, ys, 2 * NL - 1 - k);
    MSM_CHECK_COL_END(col);
    t.v[k - NL] = (uint32_t)col & LMASK;
    col >>= LB;
  }
  MSM_CHECK(col < (1ull << 28));
  t.v[NL - 1] = (uint32_t)col;
  r = t;
}

// r = a*a*R^-1 (mod p), class M.  Same columns as fe_mul, but each cross product a_i*a_j (i < j) is formed once
// against the pre-doubled limb 2*a_j: 105 multiplies instead of 196 for the product half.
// Bound: limbs < 2^30  =>  7*2^30*2^31 + 2^60 + 14*(2^28)^2 + carry < 2^64.
template <class F>
MSM_HD void fe_sqr(Fe& r, const Fe& a, const Modulus<F>& md) {
  if constexpr (F::N != NL) {   // 13 x 29: no dedicated squaring (the twisted-Edwards law has none on its hot path)
    fe_mul<F>(r, a, a, md);
    return;
  } else {
  uint32_t m[NL], a2[NL], xs[NL], ys[NL];
  Fe t;
  uint64_t col = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    MSM_CHECK(a.v[i] < (1u << 30));
    a2[i] = a.v[i] << 1;
  }
#pragma unroll
  for (int k = 0; k < NL; k++) {
    MSM_CHECK_COL_BEGIN();
    int n = 0;
#pragma unroll
    for (int i = 0; 2 * i < k; i++) {
      xs[n] = a.v[i];
      ys[n] = a2[k - i];
      n++;
      MSM_CHECK_COL_ADD((unsigned __int128)a.v[i] * a2[k - i]);
    }
    if ((k & 1) == 0) {
      xs[n] = a.v[k / 2];
      ys[n] = a.v[k / 2];
      n++;
      MSM_CHECK_COL_ADD((unsigned __int128)a.v[k / 2] * a.v[k / 2]);
    }
    mad_chain<false>(col, xs, ys, n);
#pragma unroll
    for (int i = 0; i < k; i++) {
      xs[i] = m[i];
      ys[i] = md.p[k - i];
      MSM_CHECK_COL_ADD((unsigned __int128)m[i] * md.p[k - i]);
    }
    mad_chain<true>(col, xs, ys, k);
    MSM_MONT_STEP(F, col, m[k], md);
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
    MSM_CHECK_COL_BEGIN();
    int n = 0;
#pragma unroll
    for (int i = k - NL + 1; 2 * i < k; i++) {
      xs[n] = a.v[i];
      ys[n] = a2[k - i];
      n++;
      MSM_CHECK_COL_ADD((unsigned __int128)a.v[i] * a2[k - i]);
    }
    if ((k & 1) == 0) {
      xs[n] = a.v[k / 2];
      ys[n] = a.v[k / 2];
      n++;
      MSM_CHECK_COL_ADD((unsigned __int128)a.v[k / 2] * a.v[k / 2]);
    }
    mad_chain<false>(col, xs, ys, n);
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) {
      xs[i - (k - NL + 1)] = m[i];
      ys[i - (k - NL + 1)] = md.p[k - i];
      MSM_CHECK_COL_ADD((unsigned __int128)m[i] * md.p[k - i]);
    }
    mad_chain<true>(col, xs, ys, 2 * NL - 1 - k);
    MSM_CHECK_COL_END(col);
    t.v[k - NL] = (uint32_t)col & LMASK;
    col >>= LB;
  }
  t.v[NL - 1] = (uint32_t)col;
  r = t;
  }
}

// Limb-wise r = a + b.  Caller tracks bounds (limbs add, values add).  N = limbs in use (words N.. of the result are 0).
template <int N = NL>
MSM_HD void fe_add(Fe& r, const Fe& a, const Fe& b) {
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = a.v[i] + b.v[i];
#pragma unroll
  for (int i = N; i < NL; i++) r.v[i] = 0;
}

template <int N = NL>
MSM_HD void fe_dbl(Fe& r, const Fe& a) {
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = a.v[i] << 1;
#pragma unroll
  for (int i = N; i < NL; i++) r.v[i] = 0;
}

// Limb-wise r = a + (k*p lifted) - b.  `bias` is one of F::BIASk_l: k*p with each limb raised by
// 2^l, so no limb underflows when b's limbs are <= 2^l and value(b) <= k*p.  value(r) = a + k*p - b.
template <int N = NL>
MSM_HD void fe_sub(Fe& r, const Fe& a, const Fe& b, const uint32_t (&bias)[NL]) {
#pragma unroll
  for (int i = 0; i < N; i++) {
    MSM_CHECK(bias[i] >= b.v[i]);
    MSM_CHECK((uint64_t)a.v[i] + bias[i] - b.v[i] < (1ull << 32));
    r.v[i] = a.v[i] + (bias[i] - b.v[i]);
  }
#pragma unroll
  for (int i = N; i < NL; i++) r.v[i] = 0;
}

// r = (k*p) - b  (negation, value in (0, k*p]).
template <int N = NL>
MSM_HD void fe_neg(Fe& r, const Fe& b, const uint32_t (&bias)[NL]) {
#pragma unroll
  for (int i = 0; i < N; i++) {
    MSM_CHECK(bias[i] >= b.v[i]);
    r.v[i] = bias[i] - b.v[i];
  }
#pragma unroll
  for (int i = N; i < NL; i++) r.v[i] = 0;
}

// One parallel carry pass: every limb keeps its low B bits and receives its lower neighbour's
// overflow.  Limbs up to 2^32-1 come out < 2^B + 2^(32-B); the top limb only receives; the value is unchanged.
template <int N = NL, int B = LB>
MSM_HD void fe_carry(Fe& r) {
  constexpr uint32_t MASK = (1u << B) - 1;
  uint32_t c[N - 1];
#pragma unroll
  for (int i = 0; i < N - 1; i++) c[i] = r.v[i] >> B;
  r.v[0] &= MASK;
#pragma unroll
  for (int i = 1; i < N - 1; i++) r.v[i] = (r.v[i] & MASK) + c[i - 1];
  r.v[N - 1] += c[N - 2];
}

// Sequential carry propagation to strictly normalized limbs (v[0..N-2] < 2^B).
template <int N = NL, int B = LB>
MSM_HD void fe_normalize(Fe& r) {
  constexpr uint32_t MASK = (1u << B) - 1;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < N - 1; i++) {
    uint32_t t = r.v[i] + c;
    r.v[i] = t & MASK;
    c = t >> B;
  }
  r.v[N - 1] += c;
}

// Cheap partial reduction for lazy values: in  value < 32p, limbs < 2^31;  out  value < 3p, strictly normalized.
// q = floor(top_limb / (p_top + 1)) under-estimates floor(value / p) by at most 2 (the multiply-shift division adds
// one more unit of slack), so value - q*p stays non-negative and below 3p.  ~110 cheap ops, no multiplier chain.
template <class F>
MSM_HD void fe_weak_reduce(Fe& r) {
  static_assert(F::N == NL && F::B == LB, "fe_weak_reduce is written for the 14 x 28 shape");
  fe_normalize(r);
  constexpr uint32_t PT = F::P[NL - 1];
  constexpr uint32_t MAGIC = (uint32_t)((1ull << 32) / (PT + 1));
  const uint32_t q = (uint32_t)(((uint64_t)r.v[NL - 1] * MAGIC) >> 32);
  MSM_CHECK(q < 64);
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    c += (int64_t)r.v[i] - (int64_t)((uint64_t)q * F::P[i]);
    if (i < NL - 1) {
      r.v[i] = (uint32_t)c & LMASK;
      c >>= LB;
    } else {
      MSM_CHECK(c >= 0 && c < (1ll << 28));
      r.v[i] = (uint32_t)c;
    }
  }
}

// a >= b on strictly normalized limbs.
template <int N = NL>
MSM_HD bool fe_geq(const Fe& a, const uint32_t (&b)[NL]) {
  bool ge = true;  // equal so far => a >= b
#pragma unroll
  for (int i = 0; i < N; i++) {
    if (a.v[i] != b[i]) ge = a.v[i] > b[i];
  }
  return ge;
}

// Canonical representative in [0, p), strictly normalized.  Input: limbs < 2^31, value < 64p (and < 2^(B(N-1)+32): 9p for the
// 13 x 29 shape).  Slow path (zero tests in rare branches, final output); the hot loop never calls it.
template <class F>
MSM_HD void fe_reduce(Fe& r) {
  constexpr int N = F::N, B = F::B;
  constexpr uint32_t MASK = (1u << B) - 1;
  fe_normalize<N, B>(r);
#pragma unroll 1
  for (int k = (N == NL ? 32 : 8); k >= 1; k >>= 1) {
    // subtract k*p if r >= k*p
    uint32_t kp[NL];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      if (i < N) {
        c += (uint64_t)F::P[i] * (uint32_t)k;
        kp[i] = (i == N - 1) ? (uint32_t)c : ((uint32_t)c & MASK);
        c >>= B;
      } else {
        kp[i] = 0;
      }
    }
    if (fe_geq<N>(r, kp)) {
      int32_t borrow = 0;
#pragma unroll
      for (int i = 0; i < N; i++) {
        int64_t d = (int64_t)r.v[i] - kp[i] + borrow;
        if (i < N - 1) {
          r.v[i] = (uint32_t)d & MASK;
          borrow = (int32_t)(d >> B);
        } else {
          r.v[i] = (uint32_t)d;
        }
      }
    }
  }
}

// value == 0 (mod p) for a class-M element (normalized, < 2p): it is 0 or p.  The first-limb test
// rejects all but ~2^-27 of the non-zero values, so the full compare is off the hot path.
template <class F>
MSM_HD bool fe_is_zero_M(const Fe& a) {
  if (a.v[0] != 0 && a.v[0] != F::P[0]) return false;
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < F::N; i++) {
    z |= a.v[i];
    e |= a.v[i] ^ F::P[i];
  }
  return z == 0 || e == 0;
}

// General zero test (any bounded lazy value).
template <class F>
MSM_HD bool fe_is_zero_slow(const Fe& a) {
  Fe t = a;
  fe_reduce<F>(t);
  uint32_t z = 0;
#pragma unroll
  for (int i = 0; i < F::N; i++) z |= t.v[i];
  return z == 0;
}

// Lane mask of a per-lane condition, in an SGPR pair: the form v_cndmask_b32_e64 takes.
// Why not leave the select to hipcc: it emits the VOP2 encoding (v_cndmask_b32_e32, condition implicit in VCC), which
// gfx950 issues at 22.9 cycles per wave-instruction against 4.2 for the VOP3 encoding with the mask in VCC or any SGPR
// pair (tools/ubench_valu.hip, profiles/r02_ubench_valu_w4.txt) -- 42 such selects were 7 % of a mixed addition.
struct LaneMask {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t m;
#else
  bool m;
#endif
};
MSM_HD LaneMask lane_mask(bool take) {
#if defined(__HIP_DEVICE_COMPILE__)
  return LaneMask{__builtin_amdgcn_ballot_w64(take)};
#else
  return LaneMask{take};
#endif
}
MSM_HD uint32_t sel_u32(uint32_t if_clear, uint32_t if_set, const LaneMask& k) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(MSM_CMOV_PLAIN)
  return ((k.m >> __lane_id()) & 1) ? if_set : if_clear;   // A/B only: lets hipcc pick v_cndmask_b32_e32 again
#elif defined(__HIP_DEVICE_COMPILE__)
  uint32_t d;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(if_clear), "v"(if_set), "s"(k.m));
  return d;
#else
  return k.m ? if_set : if_clear;
#endif
}

template <int N = NL>
MSM_HD void fe_cmov(Fe& r, const Fe& a, const LaneMask& k) {
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = sel_u32(r.v[i], a.v[i], k);
}
template <int N = NL>
MSM_HD void fe_cmov(Fe& r, const Fe& a, bool take) { fe_cmov<N>(r, a, lane_mask(take)); }

// ---- ABI conversions (6 x u64 little-endian, Montgomery radix 2^384  <->  internal) ----------

// 48 bytes (12 u32 words, little-endian) -> radix-2^B limbs of the same integer (the top limb takes what is left: for 13 x 29 the
// value must stay below 2^380).
template <int N = NL, int B = LB>
MSM_HD void fe_from_words(Fe& r, const uint32_t* w) {
  constexpr uint32_t MASK = (1u << B) - 1;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    if (i >= N) {
      r.v[i] = 0;
      continue;
    }
    int bit = B * i;
    int wi = bit >> 5, sh = bit & 31;
    uint64_t lo = (wi < 12) ? w[wi] : 0;
    uint64_t hi = (wi + 1 < 12) ? w[wi + 1] : 0;
    const uint32_t x = (uint32_t)(((lo | (hi << 32)) >> sh));
    r.v[i] = (i == N - 1 && N * B < 384) ? x : (x & MASK);
  }
}

// strictly normalized limbs with value < 2^384 -> 12 u32 words.
template <int N = NL, int B = LB>
MSM_HD void fe_to_words(uint32_t* w, const Fe& a) {
#pragma unroll
  for (int j = 0; j < 12; j++) {
    int bit = 32 * j;
    int li = bit / B, sh = bit % B;
    uint64_t acc = li < N ? (uint64_t)a.v[li] >> sh : 0;
    int have = B - sh;
    if (li + 1 < N) acc |= (uint64_t)a.v[li + 1] << have;
    have += B;
    if (have < 32 && li + 2 < N) acc |= (uint64_t)a.v[li + 2] << have;
    w[j] = (uint32_t)acc;
  }
}

// ABI Montgomery (x*2^384 mod p, canonical) -> internal class M (x*2^392 mod p).
template <class F>
MSM_HD void fe_from_abi(Fe& r, const uint32_t* w, const Modulus<F>& md) {
  static_assert(F::N == NL && F::B == LB, "the ABI boundary is crossed in the 14 x 28 shape");
  Fe t, c;
  fe_from_words(t, w);
  fe_set(c, F::CIN);
  fe_mul<F>(r, t, c, md);
}

// internal lazy value (limbs < 2^30, value < 32p) -> canonical ABI Montgomery words.
template <class F>
MSM_HD void fe_to_abi(uint32_t* w, const Fe& a, const Modulus<F>& md) {
  Fe t, c;
  fe_set(c, F::COUT);
  fe_mul<F>(t, a, c, md);
  fe_reduce<F>(t);
  fe_to_words(w, t);
}

// ---- between the two shapes of BLS12-377 Fq (same residue, Montgomery radix 2^392 <-> 2^406) ---------------------------------
// in: canonical (fe_reduce'd) 14 x 28 value x*2^392;  out: class M 13 x 29 value x*2^406
MSM_HD void fe_28_to_29(Fe& r, const Fe& a, const Modulus<Bls12_377_Fq29>& md29) {
  uint32_t w[12];
  fe_to_words(w, a);
  Fe t, c;
  fe_from_words<Bls12_377_Fq29::N, Bls12_377_Fq29::B>(t, w);
  fe_set(c, Bls12_377_Fq29::FROM28);
  fe_mul<Bls12_377_Fq29>(r, t, c, md29);
}
// in: any bounded 13 x 29 value (< 9p) x*2^406;  out: class M 14 x 28 value x*2^392
MSM_HD void fe_29_to_28(Fe& r, const Fe& a, const Modulus<Bls12_377_Fq>& md28) {
  Fe t = a, c;
  fe_reduce<Bls12_377_Fq29>(t);
  uint32_t w[12];
  fe_to_words<Bls12_377_Fq29::N, Bls12_377_Fq29::B>(w, t);
  fe_from_words(t, w);
  fe_set(c, Bls12_377_Cross::TO28);
  fe_mul<Bls12_377_Fq>(r, t, c, md28);
}

}  // namespace msm
