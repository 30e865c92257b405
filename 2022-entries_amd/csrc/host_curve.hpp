// host_curve.hpp -- host-side use of the fp28/curve templates: ABI conversions, inversion, the final
// window fold (Horner) and the multi-GPU partial fold.  This is the part every reference entry also
// leaves on the host: SPK msm/pippenger.cuh:556-614 (accumulate), CMB yrrid-ff-ec/HostReduce.cpp:61-78,
// P1A matter-labs/src/lib.rs:32-39.  Only O(windows * window_bits) point operations happen here.
#pragma once
#include <stddef.h>
#include <string.h>

#include "curve.hpp"

namespace msm {

// arkworks Affine image (x, y in the ABI Montgomery radix, infinity flag after the two coordinates) -> internal Affine.
// The flag byte is authoritative (zero is (0,1,true) in ark 0.3 and (0,0,true) in 0.4).
template <class E>
inline bool affine_from_abi(AffineT<typename E::T>& out, const uint8_t* p, const typename E::Md& md) {
  constexpr int CB = E::WORDS * 4;
  uint32_t w[2 * E::WORDS];
  memcpy(w, p, 2 * CB);
  if (p[2 * CB] != 0) {
    E::zero(out.x);
    E::zero(out.y);
    return true;
  }
  E::from_abi(out.x, w, md);
  E::from_abi(out.y, w + E::WORDS, md);
  return false;
}

// XYZZ -> arkworks Projective image, normalised: (x, y, 1) or (1, 1, 0) for infinity
// (ARK ec/src/models/short_weierstrass.rs:750-756), all in the ABI Montgomery radix.
template <class E>
inline void xyzz_to_projective_abi(uint8_t* out, const XyzzT<typename E::T>& a, const typename E::Md& md) {
  constexpr int CB = E::WORDS * 4;
  uint32_t w[3 * E::WORDS];
  typename E::T one;
  E::set_one(one);
  if (xyzz_is_inf<E>(a)) {
    E::to_abi(w, one, md);
    E::to_abi(w + E::WORDS, one, md);
    memset(w + 2 * E::WORDS, 0, CB);
    memcpy(out, w, 3 * CB);
    return;
  }
  typename E::T t, ti, zzi, zzzi, x, y;
  E::mul(t, a.zz, a.zzz, md);
  el_inv(ti, t, md, (E*)nullptr);
  E::mul(zzi, ti, a.zzz, md);
  E::mul(zzzi, ti, a.zz, md);
  E::mul(x, a.x, zzi, md);
  E::mul(y, a.y, zzzi, md);
  E::to_abi(w, x, md);
  E::to_abi(w + E::WORDS, y, md);
  E::to_abi(w + 2 * E::WORDS, one, md);
  memcpy(out, w, 3 * CB);
}

// arkworks Projective (Jacobian X, Y, Z) image -> XYZZ (X, Y, Z^2, Z^3).
template <class E>
inline void xyzz_from_projective_abi(XyzzT<typename E::T>& out, const uint8_t* p, const typename E::Md& md) {
  constexpr int CB = E::WORDS * 4;
  uint32_t w[3 * E::WORDS];
  memcpy(w, p, 3 * CB);
  typename E::T z;
  E::from_abi(out.x, w, md);
  E::from_abi(out.y, w + E::WORDS, md);
  E::from_abi(z, w + 2 * E::WORDS, md);
  E::sqr(out.zz, z, md);
  E::mul(out.zzz, out.zz, z, md);
}

// result = sum_w 2^(c*w) * sums[w]   (window combine, high to low).
template <class E>
inline void fold_windows(XyzzT<typename E::T>& acc, const XyzzT<typename E::T>* sums, int windows, int c, const typename E::Md& md) {
  xyzz_set_inf<E>(acc);
  for (int w = windows - 1; w >= 0; w--) {
    if (!xyzz_is_inf<E>(acc)) {
      for (int i = 0; i < c; i++) xyzz_dbl<E>(acc, md);
    }
    xyzz_add<E>(acc, sums[w], md);
  }
}

// ---- curve descriptors: coordinate field policy, scalar field, ABI sizes, generator -----------------------------
struct Bls12_377_G1 {
  using E = FpEl<Bls12_377_Fq>;
  using FR = Bls12_377_Fr;
  static constexpr int SCALAR_BITS = 253;
  static void generator(AffineT<Fe>& g) { fe_set(g.x, Bls12_377_Fq::G1X); fe_set(g.y, Bls12_377_Fq::G1Y); }
};
struct Bls12_381_G1 {
  using E = FpEl<Bls12_381_Fq>;
  using FR = Bls12_381_Fr;
  static constexpr int SCALAR_BITS = 255;
  static void generator(AffineT<Fe>& g) { fe_set(g.x, Bls12_381_Fq::G1X); fe_set(g.y, Bls12_381_Fq::G1Y); }
};
struct Bls12_377_G2 {
  using E = Fp2El<Bls12_377_Fq, 5>;
  using FR = Bls12_377_Fr;
  static constexpr int SCALAR_BITS = 253;
  static void generator(AffineT<Fe2>& g) {
    fe_set(g.x.c0, Bls12_377_Fq::G2X0); fe_set(g.x.c1, Bls12_377_Fq::G2X1);
    fe_set(g.y.c0, Bls12_377_Fq::G2Y0); fe_set(g.y.c1, Bls12_377_Fq::G2Y1);
  }
};
struct Bls12_381_G2 {   // Fq2 = Fq[u]/(u^2 + 1): ARKC bls12_381/src/fields/fq2.rs:13; generator curves/g2.rs:74-91
  using E = Fp2El<Bls12_381_Fq, 1>;
  using FR = Bls12_381_Fr;
  static constexpr int SCALAR_BITS = 255;
  static void generator(AffineT<Fe2>& g) {
    fe_set(g.x.c0, Bls12_381_Fq::G2X0); fe_set(g.x.c1, Bls12_381_Fq::G2X1);
    fe_set(g.y.c0, Bls12_381_Fq::G2Y0); fe_set(g.y.c1, Bls12_381_Fq::G2Y1);
  }
};

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

template <class E>
inline void xyzz_mul_u64x4(XyzzT<typename E::T>& r, const AffineT<typename E::T>& g, const uint64_t k[4], const typename E::Md& md) {
  xyzz_set_inf<E>(r);
  for (int bit = 255; bit >= 0; bit--) {
    if (!xyzz_is_inf<E>(r)) xyzz_dbl<E>(r, md);
    if ((k[bit >> 6] >> (bit & 63)) & 1) xyzz_madd<E>(r, g, false, false, md);
  }
}

template <class C>
inline void generate_points(uint64_t seed, size_t distinct, size_t npoints, uint8_t* out, size_t stride) {
  using E = typename C::E;
  using El = typename E::T;
  typename E::Md md;
  if (distinct > npoints) distinct = npoints;
  if (distinct == 0) return;
  AffineT<El> g;
  C::generator(g);
  uint64_t st = seed, h0[4], h1[4];
  for (int i = 0; i < 4; i++) h0[i] = splitmix64(st);
  for (int i = 0; i < 4; i++) h1[i] = splitmix64(st);
  h0[3] &= 0x03ffffffffffffffull;  // 250-bit multipliers: below both group orders
  h1[3] &= 0x03ffffffffffffffull;
  h1[0] |= 1;
  XyzzT<El> acc, step;
  xyzz_mul_u64x4<E>(acc, g, h0, md);
  xyzz_mul_u64x4<E>(step, g, h1, md);
  XyzzT<El>* pts = new XyzzT<El>[distinct];
  El* prefix = new El[distinct];
  El run;
  E::set_one(run);
  for (size_t j = 0; j < distinct; j++) {
    if (xyzz_is_inf<E>(acc)) xyzz_add<E>(acc, step, md);  // (measure-zero) skip the identity
    pts[j] = acc;
    prefix[j] = run;                       // product of zz*zzz of all earlier points
    El t;
    E::mul(t, acc.zz, acc.zzz, md);
    E::mul(run, run, t, md);
    xyzz_add<E>(acc, step, md);
  }
  El inv;
  el_inv(inv, run, md, (E*)nullptr);
  constexpr int CB = E::WORDS * 4;
  for (size_t j = distinct; j-- > 0;) {
    El t, ti, zzi, zzzi, x, y;
    E::mul(ti, inv, prefix[j], md);        // (zz_j*zzz_j)^-1
    E::mul(t, pts[j].zz, pts[j].zzz, md);
    E::mul(inv, inv, t, md);
    E::mul(zzi, ti, pts[j].zzz, md);
    E::mul(zzzi, ti, pts[j].zz, md);
    E::mul(x, pts[j].x, zzi, md);
    E::mul(y, pts[j].y, zzzi, md);
    uint32_t w[2 * E::WORDS];
    E::to_abi(w, x, md);
    E::to_abi(w + E::WORDS, y, md);
    uint8_t* o = out + j * stride;
    memset(o, 0, stride);
    memcpy(o, w, 2 * CB);
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
