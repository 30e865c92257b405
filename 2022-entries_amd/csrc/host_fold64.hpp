// host_fold64.hpp -- the host tail of an MSM (window combine, chunk sums, normalisation) on 6 x 64-bit limbs.
//
// The device representation (fp28.hpp: 14 limbs of 28 bits, lazily reduced) is built for the GPU's 32x32 multiplier; run on a
// CPU core it costs ~150 ns per field multiplication, and the ~260 sequential doublings of the Horner fold plus one
// inversion were 0.4-0.5 ms per MSM -- half the wall time of a small MSM.  The same arithmetic on 64-bit limbs (CIOS with
// 128-bit products, fully reduced) is ~4x faster.  This is the part every reference entry also leaves on the host
// (SPK msm/pippenger.cuh:556-614, CMB yrrid-ff-ec/HostReduce.cpp:61-78); the formulas are the public EFD ones used in
// curve.hpp / te.hpp, re-stated for canonical values.  The short-Weierstrass functions are generic over the coordinate
// field: Fp64 for G1, Fp2_64 (Fp2 over it) for G2 -- whose fold of 37 windows on the device representation was 1.7 ms, as long
// as the whole device side of a small G2 MSM.
//
// Values are Montgomery residues with R = 2^384 -- exactly the ABI's representation, so results need no conversion.
#pragma once
#include <stdint.h>
#include <string.h>

#include "curve.hpp"

namespace msm {

struct F64 {
  uint64_t l[6];
};

#if defined(__x86_64__)
// One Montgomery product on BMI2 + ADX (mulx with two independent carry chains, adcx / adox): ~2x the code clang makes of the
// portable loop below.  The row structure is that of Fp64::mul (no-carry CIOS); the seven accumulator words rotate through
// r8..r14 so that the shift by one word per row is a renaming.  t = a b R^-1 mod p up to one subtraction of p (the caller does it).
#define MSM_MONT_ROW(i, t0, t1, t2, t3, t4, t5, t6)                                                   \
  "movq " #i "*8(%[b]), %%rdx\n\t"                                                                    \
  "xorq %%rax, %%rax\n\t"                                                                             \
  "mulxq 0(%[a]), %%rax, %%rbx\n\t adcxq %%rax, %%" #t0 "\n\t adoxq %%rbx, %%" #t1 "\n\t"             \
  "mulxq 8(%[a]), %%rax, %%rbx\n\t adcxq %%rax, %%" #t1 "\n\t adoxq %%rbx, %%" #t2 "\n\t"             \
  "mulxq 16(%[a]), %%rax, %%rbx\n\t adcxq %%rax, %%" #t2 "\n\t adoxq %%rbx, %%" #t3 "\n\t"            \
  "mulxq 24(%[a]), %%rax, %%rbx\n\t adcxq %%rax, %%" #t3 "\n\t adoxq %%rbx, %%" #t4 "\n\t"            \
  "mulxq 32(%[a]), %%rax, %%rbx\n\t adcxq %%rax, %%" #t4 "\n\t adoxq %%rbx, %%" #t5 "\n\t"            \
  "mulxq 40(%[a]), %%rax, %%rbx\n\t adcxq %%rax, %%" #t5 "\n\t adoxq %%rbx, %%" #t6 "\n\t"            \
  "adcxq %[zero], %%" #t6 "\n\t"                                                                      \
  "movq %%" #t0 ", %%rdx\n\t imulq %[inv], %%rdx\n\t"                                                 \
  "xorq %%rax, %%rax\n\t"                                                                             \
  "mulxq 0(%[p]), %%rax, %%rbx\n\t adcxq %%rax, %%" #t0 "\n\t adoxq %%rbx, %%" #t1 "\n\t"             \
  "mulxq 8(%[p]), %%rax, %%rbx\n\t adcxq %%rax, %%" #t1 "\n\t adoxq %%rbx, %%" #t2 "\n\t"             \
  "mulxq 16(%[p]), %%rax, %%rbx\n\t adcxq %%rax, %%" #t2 "\n\t adoxq %%rbx, %%" #t3 "\n\t"            \
  "mulxq 24(%[p]), %%rax, %%rbx\n\t adcxq %%rax, %%" #t3 "\n\t adoxq %%rbx, %%" #t4 "\n\t"            \
  "mulxq 32(%[p]), %%rax, %%rbx\n\t adcxq %%rax, %%" #t4 "\n\t adoxq %%rbx, %%" #t5 "\n\t"            \
  "mulxq 40(%[p]), %%rax, %%rbx\n\t adcxq %%rax, %%" #t5 "\n\t adoxq %%rbx, %%" #t6 "\n\t"            \
  "adcxq %[zero], %%" #t6 "\n\t"

// (under AddressSanitizer locals live on a fake stack addressed through one more register than this block can spare -- it
//  clobbers ten and takes four pointers: the function itself is left uninstrumented there, its callers are not)
#if defined(__SANITIZE_ADDRESS__)
__attribute__((no_sanitize_address, noinline))
#endif
inline void mont_mul_adx(uint64_t* out, const uint64_t* a, const uint64_t* b, const uint64_t* p, uint64_t inv) {
  static const uint64_t zero = 0;
  asm volatile(
      "xorq %%r8, %%r8\n\t xorq %%r9, %%r9\n\t xorq %%r10, %%r10\n\t xorq %%r11, %%r11\n\t xorq %%r12, %%r12\n\t xorq %%r13, %%r13\n\t"
      "xorq %%r14, %%r14\n\t"
      MSM_MONT_ROW(0, r8, r9, r10, r11, r12, r13, r14)
      MSM_MONT_ROW(1, r9, r10, r11, r12, r13, r14, r8)
      MSM_MONT_ROW(2, r10, r11, r12, r13, r14, r8, r9)
      MSM_MONT_ROW(3, r11, r12, r13, r14, r8, r9, r10)
      MSM_MONT_ROW(4, r12, r13, r14, r8, r9, r10, r11)
      MSM_MONT_ROW(5, r13, r14, r8, r9, r10, r11, r12)
      "movq %%r14, 0(%[out])\n\t movq %%r8, 8(%[out])\n\t movq %%r9, 16(%[out])\n\t movq %%r10, 24(%[out])\n\t movq %%r11, 32(%[out])\n\t"
      "movq %%r12, 40(%[out])\n\t"
      :
      : [out] "r"(out), [a] "r"(a), [b] "r"(b), [p] "r"(p), [inv] "m"(inv), [zero] "m"(zero)
      : "rax", "rbx", "rdx", "r8", "r9", "r10", "r11", "r12", "r13", "r14", "cc", "memory");
}
#undef MSM_MONT_ROW
inline bool cpu_has_mulx_adx() { return __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx"); }
#else
inline void mont_mul_adx(uint64_t*, const uint64_t*, const uint64_t*, const uint64_t*, uint64_t) {}
inline bool cpu_has_mulx_adx() { return false; }
#endif

struct Fp64 {
  using El = F64;
  static constexpr int COORD_BYTES = 48;
  uint64_t p[6];
  uint64_t inv;      // -p^-1 mod 2^64
  bool adx = false;  // mulx / adcx / adox present: mul() takes the assembly path
  F64 one;           // R mod p
  F64 from28;        // 2^376 mod p: mont_mul(v, from28) turns v = x * 2^392 (device radix) into x * 2^384
  F64 from29;        // 2^362 mod p: the same for the 13 x 29 shape of BLS12-377 (v = x * 2^406; fp28.hpp)
  F64 two_d;         // twisted-Edwards 2d (BLS12-377 only), Montgomery form
  F64 sqrt3, fsc_sqrt3;   // constants of the map back to short Weierstrass (te.hpp::te_to_sw)

  static bool geq(const uint64_t* a, const uint64_t* b) {
    for (int i = 5; i >= 0; i--)
      if (a[i] != b[i]) return a[i] > b[i];
    return true;
  }
  void cond_sub(uint64_t* t, uint64_t top) const {
    if (top || geq(t, p)) {
      unsigned __int128 br = 0;
      for (int i = 0; i < 6; i++) {
        const unsigned __int128 d = (unsigned __int128)t[i] - p[i] - (uint64_t)br;
        t[i] = (uint64_t)d;
        br = (d >> 64) & 1;
      }
    }
  }
  void add(F64& r, const F64& a, const F64& b) const {
    unsigned __int128 c = 0;
    uint64_t t[6];
    for (int i = 0; i < 6; i++) {
      c += (unsigned __int128)a.l[i] + b.l[i];
      t[i] = (uint64_t)c;
      c >>= 64;
    }
    cond_sub(t, (uint64_t)c);
    memcpy(r.l, t, sizeof t);
  }
  void sub(F64& r, const F64& a, const F64& b) const {
    unsigned __int128 br = 0;
    uint64_t t[6];
    for (int i = 0; i < 6; i++) {
      const unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - (uint64_t)br;
      t[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
    if (br) {
      unsigned __int128 c = 0;
      for (int i = 0; i < 6; i++) {
        c += (unsigned __int128)t[i] + p[i];
        t[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    memcpy(r.l, t, sizeof t);
  }
  void dbl(F64& r, const F64& a) const { add(r, a, a); }
  void neg(F64& r, const F64& a) const {
    F64 z{};
    sub(r, z, a);
  }
  bool is_zero(const F64& a) const { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }
  bool eq(const F64& a, const F64& b) const { return memcmp(a.l, b.l, sizeof a.l) == 0; }

  // Montgomery multiplication, r = a b R^-1 mod p, canonical: CIOS with the two passes of a row fused into one loop and no
  // carry word above the sixth limb -- the "no-carry" form arkworks uses when the modulus leaves its top bit free
  // (ARK ff montgomery_backend.rs:146-201, CAN_USE_NO_CARRY_MUL_OPT), which both base fields do (377 and 381 bits of 384).
  // Two independent 64-bit carry chains per row (A: a_j b_i, C: m p_j); 28 -> 19 ns on the GPU node's EPYC, and the fold of
  // a small MSM is ~2500 of these.
  void mul(F64& r, const F64& a, const F64& b) const {
    if (adx) {
      uint64_t t[6];
      mont_mul_adx(t, a.l, b.l, p, inv);
      cond_sub(t, 0);
      memcpy(r.l, t, 48);
      return;
    }
    mul_portable(r, a, b);
  }
  void mul_portable(F64& r, const F64& a, const F64& b) const {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
      unsigned __int128 A = (unsigned __int128)a.l[0] * b.l[i] + t[0];
      const uint64_t m = (uint64_t)A * inv;
      unsigned __int128 C = (unsigned __int128)m * p[0] + (uint64_t)A;
      A >>= 64;
      C >>= 64;
      for (int j = 1; j < 6; j++) {
        A += (unsigned __int128)a.l[j] * b.l[i] + t[j];
        C += (unsigned __int128)m * p[j] + (uint64_t)A;
        t[j - 1] = (uint64_t)C;
        A >>= 64;
        C >>= 64;
      }
      t[5] = (uint64_t)C + (uint64_t)A;   // no overflow: p < 2^383
    }
    cond_sub(t, 0);
    memcpy(r.l, t, 48);
  }
  void sqr(F64& r, const F64& a) const { mul(r, a, a); }
  // a^(p-2)
  void invert(F64& r, const F64& a) const {
    uint64_t e[6];
    memcpy(e, p, sizeof e);
    e[0] -= 2;   // p is odd and > 2: no borrow
    F64 acc = one;
    for (int i = 5; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        sqr(acc, acc);
        if ((e[i] >> b) & 1) mul(acc, acc, a);
      }
    r = acc;
  }

  const F64& one_el() const { return one; }
  void store(uint8_t* out, const F64& a) const { memcpy(out, a.l, 48); }
  // device field element (any bounded lazy value, in the limb shape of F) -> F64
  template <class F>
  void from_device(F64& r, const Fe& a) const {
    Fe t = a;
    fe_reduce<F>(t);
    uint32_t w[12];
    fe_to_words<F::N, F::B>(w, t);
    F64 v;
    for (int i = 0; i < 6; i++) v.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    mul(r, v, F::B == 29 ? from29 : from28);
  }
  void from_limbs28(F64& r, const uint32_t (&c)[NL]) const {   // a plain integer given as radix-2^28 limbs (< p)
    Fe t;
    fe_set(t, c);
    uint32_t w[12];
    fe_to_words(w, t);
    for (int i = 0; i < 6; i++) r.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
  }

  template <class F>
  void init() {
    Fe pp;
    fe_set(pp, F::P);
    uint32_t w[12];
    fe_to_words(w, pp);
    for (int i = 0; i < 6; i++) p[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    uint64_t x = 1;   // Newton: x <- x (2 - p0 x) doubles the correct low bits
    for (int i = 0; i < 7; i++) x *= 2 - p[0] * x;
    inv = 0 - x;
    adx = cpu_has_mulx_adx();
    // 2^k mod p by repeated doubling of 1
    F64 v{};
    v.l[0] = 1;
    F64 p362{}, p376{}, p384{}, p768{};
    for (int k = 1; k <= 768; k++) {
      add(v, v, v);
      if (k == 362) p362 = v;
      if (k == 376) p376 = v;
      if (k == 384) p384 = v;
      if (k == 768) p768 = v;
    }
    one = p384;
    // from28 must be the Montgomery image of 2^-8, i.e. 2^-8 * 2^384 = 2^376
    from28 = p376;
    // ... and of 2^-22 for values in Montgomery radix 2^406
    from29 = p362;
    (void)p768;
  }
  // Montgomery image of a constant stored in the device representation (x * 2^392 mod p as radix-2^28 limbs)
  template <class F>
  void const_from_device(F64& r, const uint32_t (&c)[NL]) const {
    Fe t;
    fe_set(t, c);
    from_device<F>(r, t);
  }
};

// Fp2 = Fp[u] / (u^2 + NEG_BETA) over an Fp64 (NEG_BETA = 5 for BLS12-377: ARKC bls12_377/src/fields/fq2.rs:13), ABI image c0 | c1.
struct F2_64 {
  F64 c0, c1;
};

template <int NEG_BETA>
struct Fp2_64 {
  using El = F2_64;
  static constexpr int COORD_BYTES = 96;
  const Fp64* f;
  void add(El& r, const El& a, const El& b) const { f->add(r.c0, a.c0, b.c0); f->add(r.c1, a.c1, b.c1); }
  void sub(El& r, const El& a, const El& b) const { f->sub(r.c0, a.c0, b.c0); f->sub(r.c1, a.c1, b.c1); }
  void dbl(El& r, const El& a) const { f->dbl(r.c0, a.c0); f->dbl(r.c1, a.c1); }
  bool is_zero(const El& a) const { return f->is_zero(a.c0) && f->is_zero(a.c1); }
  void times_neg_beta(F64& r, const F64& a) const {   // small constant: by additions
    F64 acc = a;
    for (int i = 1; i < NEG_BETA; i++) f->add(acc, acc, a);
    r = acc;
  }
  // (a0 + a1 u)(b0 + b1 u) = (a0 b0 - NB a1 b1) + ((a0 + a1)(b0 + b1) - a0 b0 - a1 b1) u   (quadratic_extension.rs:641-652)
  void mul(El& r, const El& a, const El& b) const {
    F64 t0, t1, t2, sa, sb;
    f->mul(t0, a.c0, b.c0);
    f->mul(t1, a.c1, b.c1);
    f->add(sa, a.c0, a.c1);
    f->add(sb, b.c0, b.c1);
    f->mul(t2, sa, sb);
    f->sub(t2, t2, t0);
    f->sub(r.c1, t2, t1);
    times_neg_beta(t1, t1);
    f->sub(r.c0, t0, t1);
  }
  void sqr(El& r, const El& a) const {
    F64 t0, t1, t2;
    f->mul(t0, a.c0, a.c0);
    f->mul(t1, a.c1, a.c1);
    f->mul(t2, a.c0, a.c1);
    f->dbl(r.c1, t2);
    times_neg_beta(t1, t1);
    f->sub(r.c0, t0, t1);
  }
  // 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + NB a1^2)
  void invert(El& r, const El& a) const {
    F64 n, t, ni;
    f->mul(n, a.c0, a.c0);
    f->mul(t, a.c1, a.c1);
    times_neg_beta(t, t);
    f->add(n, n, t);
    f->invert(ni, n);
    f->mul(r.c0, a.c0, ni);
    f->mul(t, a.c1, ni);
    f->neg(r.c1, t);
  }
  El one_el() const {
    El o{};
    o.c0 = f->one;
    return o;
  }
  void store(uint8_t* out, const El& a) const {
    memcpy(out, a.c0.l, 48);
    memcpy(out + 48, a.c1.l, 48);
  }
  template <class F>
  void from_device(El& r, const Fe2& a) const {
    f->template from_device<F>(r.c0, a.c0);
    f->template from_device<F>(r.c1, a.c1);
  }
};

template <class El>
struct XyzzG64 {
  El x, y, zz, zzz;
};
using Xyzz64 = XyzzG64<F64>;

// ---- short Weierstrass (a = 0), XYZZ: dbl-2008-s-1 and add-2008-s, canonical values; FC = Fp64 or Fp2_64 ----------------
template <class FC>
inline bool sw64_is_inf(const FC& f, const XyzzG64<typename FC::El>& a) { return f.is_zero(a.zz); }
template <class El>
inline void sw64_set_inf(XyzzG64<El>& a) { memset(&a, 0, sizeof a); }

template <class FC>
inline void sw64_dbl(const FC& f, XyzzG64<typename FC::El>& a) {
  if (sw64_is_inf(f, a)) return;
  typename FC::El u, v, w, s, m, t, x3, y3;
  f.dbl(u, a.y);
  f.sqr(v, u);
  f.mul(w, u, v);
  f.mul(s, a.x, v);
  f.sqr(m, a.x);
  f.dbl(t, m);
  f.add(m, t, m);            // 3 X^2
  f.sqr(x3, m);
  f.dbl(t, s);
  f.sub(x3, x3, t);
  f.sub(t, s, x3);
  f.mul(y3, m, t);
  f.mul(t, w, a.y);
  f.sub(y3, y3, t);
  a.x = x3;
  a.y = y3;
  f.mul(a.zz, v, a.zz);
  f.mul(a.zzz, w, a.zzz);
}

template <class FC>
inline void sw64_add(const FC& f, XyzzG64<typename FC::El>& a, const XyzzG64<typename FC::El>& b) {
  if (sw64_is_inf(f, b)) return;
  if (sw64_is_inf(f, a)) {
    a = b;
    return;
  }
  typename FC::El u1, u2, s1, s2, P, R, PP, PPP, Q, t, x3, y3;
  f.mul(u1, a.x, b.zz);
  f.mul(u2, b.x, a.zz);
  f.mul(s1, a.y, b.zzz);
  f.mul(s2, b.y, a.zzz);
  f.sub(P, u2, u1);
  f.sub(R, s2, s1);
  if (f.is_zero(P)) {
    if (f.is_zero(R))
      sw64_dbl(f, a);
    else
      sw64_set_inf(a);
    return;
  }
  f.sqr(PP, P);
  f.mul(PPP, P, PP);
  f.mul(Q, u1, PP);
  f.sqr(x3, R);
  f.sub(x3, x3, PPP);
  f.dbl(t, Q);
  f.sub(x3, x3, t);
  f.sub(t, Q, x3);
  f.mul(y3, R, t);
  f.mul(t, s1, PPP);
  f.sub(y3, y3, t);
  a.x = x3;
  a.y = y3;
  f.mul(t, a.zz, b.zz);
  f.mul(a.zz, t, PP);
  f.mul(t, a.zzz, b.zzz);
  f.mul(a.zzz, t, PPP);
}

// XYZZ -> ABI Projective image, normalised: (x, y, 1) or (1, 1, 0)
template <class FC>
inline void sw64_to_abi(const FC& f, uint8_t* out, const XyzzG64<typename FC::El>& a) {
  typename FC::El x = f.one_el(), y = f.one_el(), z{};
  if (!sw64_is_inf(f, a)) {
    typename FC::El t, ti, zzi, zzzi;
    f.mul(t, a.zz, a.zzz);
    f.invert(ti, t);
    f.mul(zzi, ti, a.zzz);
    f.mul(zzzi, ti, a.zz);
    f.mul(x, a.x, zzi);
    f.mul(y, a.y, zzzi);
    z = f.one_el();
  }
  f.store(out, x);
  f.store(out + FC::COORD_BYTES, y);
  f.store(out + 2 * FC::COORD_BYTES, z);
}

template <class F, class FC, class DevEl>
inline void xyzz64_from_device(const FC& f, XyzzG64<typename FC::El>& r, const XyzzT<DevEl>& a) {
  f.template from_device<F>(r.x, a.x);
  f.template from_device<F>(r.y, a.y);
  f.template from_device<F>(r.zz, a.zz);
  f.template from_device<F>(r.zzz, a.zzz);
}

// result = sum_w 2^(c w) sums[w]   (Horner, high to low)
template <class F, class FC, class DevEl>
inline void fold_windows64(const FC& f, XyzzG64<typename FC::El>& acc, const XyzzT<DevEl>* sums, int windows, int c) {
  sw64_set_inf(acc);
  for (int w = windows - 1; w >= 0; w--) {
    for (int i = 0; i < c; i++) sw64_dbl(f, acc);
    XyzzG64<typename FC::El> s;
    xyzz64_from_device<F>(f, s, sums[w]);
    sw64_add(f, acc, s);
  }
}

// ---- twisted Edwards a = -1, extended coordinates (x = X, y = Y, zz = Z, zzz = T), as te.hpp ----------------------------
// add-2008-hwcd-3 (unified) and the dedicated doubling dbl-2008-hwcd (4M + 4S).  false = Z3 = 0: vanishing denominator.
inline bool te64_add(const Fp64& f, Xyzz64& a, const Xyzz64& b) {
  F64 a1, a2, b1, b2, A, B, C, D, E, Fv, G, H, t;
  f.sub(a1, a.y, a.x);
  f.sub(a2, b.y, b.x);
  f.add(b1, a.y, a.x);
  f.add(b2, b.y, b.x);
  f.mul(A, a1, a2);
  f.mul(B, b1, b2);
  f.mul(t, b.zzz, f.two_d);
  f.mul(C, a.zzz, t);
  f.mul(t, a.zz, b.zz);
  f.dbl(D, t);
  f.sub(E, B, A);
  f.sub(Fv, D, C);
  f.add(G, D, C);
  f.add(H, B, A);
  f.mul(a.x, E, Fv);
  f.mul(a.y, G, H);
  f.mul(a.zzz, E, H);
  f.mul(a.zz, Fv, G);
  return !f.is_zero(a.zz);
}
inline bool te64_dbl(const Fp64& f, Xyzz64& a) {
  F64 A, B, C, D, E, G, Fv, H, t;
  f.sqr(A, a.x);
  f.sqr(B, a.y);
  f.sqr(t, a.zz);
  f.dbl(C, t);
  f.neg(D, A);                 // a = -1
  f.add(t, a.x, a.y);
  f.sqr(E, t);
  f.sub(E, E, A);
  f.sub(E, E, B);
  f.add(G, D, B);
  f.sub(Fv, G, C);
  f.sub(H, D, B);
  f.mul(a.x, E, Fv);
  f.mul(a.y, G, H);
  f.mul(a.zzz, E, H);
  f.mul(a.zz, Fv, G);
  return !f.is_zero(a.zz);
}

// Horner on the Edwards image, then back to short Weierstrass (te.hpp::te_to_sw), as an XYZZ64 affine point.
// false: a vanishing denominator was hit (possible only off the odd-order subgroup) -- the caller repeats on XYZZ.
// FD = the field constants whose limb shape the device sums are in (Bls12_377_Fq29 when the Edwards kernels run on 13 x 29 limbs).
template <class F, class FD = F>
inline bool fold_windows_te64(const Fp64& f, Xyzz64& out, const Xyzz* sums, int windows, int c) {
  Xyzz64 acc{};
  acc.y = f.one;
  acc.zz = f.one;   // identity (0, 1, 1, 0)
  for (int w = windows - 1; w >= 0; w--) {
    if (w != windows - 1)
      for (int i = 0; i < c; i++)
        if (!te64_dbl(f, acc)) return false;
    Xyzz64 s;
    xyzz64_from_device<FD>(f, s, sums[w]);
    if (f.is_zero(s.zz)) return false;
    if (!te64_add(f, acc, s)) return false;
  }
  // (0, 1) -> infinity, (0, -1) -> the 2-torsion point (-1, 0); otherwise u = (Z + Y)/(Z - Y), v = FSC u Z / X
  if (f.is_zero(acc.x)) {
    if (f.eq(acc.y, acc.zz)) {
      sw64_set_inf(out);
    } else {
      f.neg(out.x, f.one);
      memset(&out.y, 0, sizeof out.y);
      out.zz = f.one;
      out.zzz = f.one;
    }
    return true;
  }
  F64 n, dn, den, inv, t;
  f.add(n, acc.zz, acc.y);
  f.sub(dn, acc.zz, acc.y);
  f.mul(den, dn, acc.x);
  f.invert(inv, den);
  f.mul(t, n, acc.x);
  f.mul(t, t, inv);                 // u
  f.mul(t, t, f.sqrt3);
  f.sub(out.x, t, f.one);
  f.mul(t, n, acc.zz);
  f.mul(t, t, inv);                 // u Z / X
  f.mul(out.y, t, f.fsc_sqrt3);
  out.zz = f.one;
  out.zzz = f.one;
  return true;
}

}  // namespace msm
