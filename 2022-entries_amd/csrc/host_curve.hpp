// host_curve.hpp -- host-side use of the fp28/curve templates: ABI conversions, inversion, the final
// window fold (Horner) and the multi-GPU partial fold.  This is the part every reference entry also
// leaves on the host: SPK msm/pippenger.cuh:556-614 (accumulate), CMB yrrid-ff-ec/HostReduce.cpp:61-78,
// P1A matter-labs/src/lib.rs:32-39.  Only O(windows * window_bits) point operations happen here.
#pragma once
#include <stddef.h>
#include <string.h>

#include "curve.cuh"

namespace msm {

// a^(p-2) by square-and-multiply over the bits of p-2 (class-M in, class-M out).
template <class F>
inline void fe_inv(Fe& r, const Fe& a, const Modulus<F>& md) {
  // exponent e = p - 2 in radix-2^28 limbs
  uint32_t e[NL];
  int64_t borrow = -2;
  for (int i = 0; i < NL; i++) {
    int64_t d = (int64_t)F::P[i] + borrow;
    if (d < 0) {
      e[i] = (uint32_t)(d + (1 << LB));
      borrow = -1;
    } else {
      e[i] = (uint32_t)d;
      borrow = 0;
    }
  }
  Fe acc;
  fe_set(acc, F::ONE);
  bool started = false;
  for (int i = NL - 1; i >= 0; i--) {
    for (int b = LB - 1; b >= 0; b--) {
      if (started) fe_sqr<F>(acc, acc, md);
      if ((e[i] >> b) & 1) {
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

// arkworks Affine image (x, y Montgomery 6xu64 LE, infinity flag at byte 96) -> internal Affine.
// The flag byte is authoritative (zero is (0,1,true) in ark 0.3 and (0,0,true) in 0.4).
template <class F>
inline bool affine_from_abi(Affine& out, const uint8_t* p, const Modulus<F>& md) {
  uint32_t w[24];
  memcpy(w, p, 96);
  if (p[96] != 0) {
    fe_zero(out.x);
    fe_zero(out.y);
    return true;
  }
  fe_from_abi<F>(out.x, w, md);
  fe_from_abi<F>(out.y, w + 12, md);
  return false;
}

// XYZZ -> arkworks Projective image, normalised: (x, y, 1) or (1, 1, 0) for infinity
// (ARK ec/src/models/short_weierstrass.rs:750-756), all in the ABI Montgomery radix.
template <class F>
inline void xyzz_to_projective_abi(uint8_t* out144, const Xyzz& a, const Modulus<F>& md) {
  uint32_t w[36];
  Fe one;
  fe_set(one, F::ONE);
  if (xyzz_is_inf<F>(a)) {
    fe_to_abi<F>(w, one, md);
    fe_to_abi<F>(w + 12, one, md);
    memset(w + 24, 0, 48);
    memcpy(out144, w, 144);
    return;
  }
  Fe t, ti, zzi, zzzi, x, y;
  fe_mul<F>(t, a.zz, a.zzz, md);
  fe_inv<F>(ti, t, md);
  fe_mul<F>(zzi, ti, a.zzz, md);
  fe_mul<F>(zzzi, ti, a.zz, md);
  fe_mul<F>(x, a.x, zzi, md);
  fe_mul<F>(y, a.y, zzzi, md);
  fe_to_abi<F>(w, x, md);
  fe_to_abi<F>(w + 12, y, md);
  fe_to_abi<F>(w + 24, one, md);
  memcpy(out144, w, 144);
}

// arkworks Projective (Jacobian X, Y, Z) image -> XYZZ (X, Y, Z^2, Z^3).
template <class F>
inline void xyzz_from_projective_abi(Xyzz& out, const uint8_t* p144, const Modulus<F>& md) {
  uint32_t w[36];
  memcpy(w, p144, 144);
  Fe z;
  fe_from_abi<F>(out.x, w, md);
  fe_from_abi<F>(out.y, w + 12, md);
  fe_from_abi<F>(z, w + 24, md);
  fe_sqr<F>(out.zz, z, md);
  fe_mul<F>(out.zzz, out.zz, z, md);
}

// result = sum_w 2^(c*w) * sums[w]   (window combine, high to low).
template <class F>
inline void fold_windows(Xyzz& acc, const Xyzz* sums, int windows, int c, const Modulus<F>& md) {
  xyzz_set_inf<F>(acc);
  for (int w = windows - 1; w >= 0; w--) {
    if (!xyzz_is_inf<F>(acc)) {
      for (int i = 0; i < c; i++) xyzz_dbl<F>(acc, md);
    }
    xyzz_add<F>(acc, sums[w], md);
  }
}

// Synthetic input generator in the shape of the reference harness (P1A yrrid/src/util.rs:15-28,
// 6block/src/util.rs:15-29): `distinct` subgroup points P_j = (h0 + j*h1) * G, batch-normalised to affine
// (one inversion, Montgomery's trick), written as arkworks Affine images and replicated by doubling the
// vector up to `npoints`.
inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

template <class F>
inline void xyzz_mul_u64x4(Xyzz& r, const Affine& g, const uint64_t k[4], const Modulus<F>& md) {
  xyzz_set_inf<F>(r);
  for (int bit = 255; bit >= 0; bit--) {
    if (!xyzz_is_inf<F>(r)) xyzz_dbl<F>(r, md);
    if ((k[bit >> 6] >> (bit & 63)) & 1) xyzz_madd<F>(r, g, false, false, md);
  }
}

template <class F>
inline void generate_points(uint64_t seed, size_t distinct, size_t npoints, uint8_t* out, size_t stride) {
  Modulus<F> md;
  if (distinct > npoints) distinct = npoints;
  if (distinct == 0) return;
  Affine g;
  fe_set(g.x, F::G1X);
  fe_set(g.y, F::G1Y);
  uint64_t st = seed, h0[4], h1[4];
  for (int i = 0; i < 4; i++) h0[i] = splitmix64(st);
  for (int i = 0; i < 4; i++) h1[i] = splitmix64(st);
  h0[3] &= 0x03ffffffffffffffull;  // 250-bit multipliers: below both group orders
  h1[3] &= 0x03ffffffffffffffull;
  h1[0] |= 1;
  Xyzz acc, step;
  xyzz_mul_u64x4<F>(acc, g, h0, md);
  xyzz_mul_u64x4<F>(step, g, h1, md);
  Xyzz* pts = new Xyzz[distinct];
  Fe* prefix = new Fe[distinct];
  Fe run;
  fe_set(run, F::ONE);
  for (size_t j = 0; j < distinct; j++) {
    if (xyzz_is_inf<F>(acc)) xyzz_add<F>(acc, step, md);  // (measure-zero) skip the identity
    pts[j] = acc;
    prefix[j] = run;                       // product of zz*zzz of all earlier points
    Fe t;
    fe_mul<F>(t, acc.zz, acc.zzz, md);
    fe_mul<F>(run, run, t, md);
    xyzz_add<F>(acc, step, md);
  }
  Fe inv;
  fe_inv<F>(inv, run, md);
  for (size_t j = distinct; j-- > 0;) {
    Fe t, ti, zzi, zzzi, x, y;
    fe_mul<F>(ti, inv, prefix[j], md);     // (zz_j*zzz_j)^-1
    fe_mul<F>(t, pts[j].zz, pts[j].zzz, md);
    fe_mul<F>(inv, inv, t, md);
    fe_mul<F>(zzi, ti, pts[j].zzz, md);
    fe_mul<F>(zzzi, ti, pts[j].zz, md);
    fe_mul<F>(x, pts[j].x, zzi, md);
    fe_mul<F>(y, pts[j].y, zzzi, md);
    uint32_t w[24];
    fe_to_abi<F>(w, x, md);
    fe_to_abi<F>(w + 12, y, md);
    uint8_t* o = out + j * stride;
    memset(o, 0, stride);
    memcpy(o, w, 96);
  }
  delete[] pts;
  delete[] prefix;
  for (size_t have = distinct; have < npoints;) {
    size_t cp = have < npoints - have ? have : npoints - have;
    memcpy(out + have * stride, out, cp * stride);
    have += cp;
  }
}

}  // namespace msm
