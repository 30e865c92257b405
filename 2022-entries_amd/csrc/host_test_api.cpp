// host_test_api.cpp -- builds libmsm_hosttest.so (plain g++, no HIP): the SAME fp28/curve templates the
// kernels use, compiled for the host with the limb-bound checker (MSM_CHECK) armed.  tests/ drive it
// through ctypes against oracle/pymodel.py.  It is test scaffolding for the arithmetic, not a product
// path: nothing in the engine or the C ABI links it.
#include <stdint.h>
#include <stdio.h>

static long g_check_failures = 0;
static char g_first_failure[256];
static void msm_check_fail(const char* file, int line, const char* cond) {
  if (g_check_failures++ == 0) snprintf(g_first_failure, sizeof g_first_failure, "%s:%d %s", file, line, cond);
}
#define MSM_CHECK(cond) do { if (!(cond)) msm_check_fail(__FILE__, __LINE__, #cond); } while (0)
#define MSM_CHECK_COL_BEGIN() unsigned __int128 chk_col_ = col
#define MSM_CHECK_COL_ADD(x) chk_col_ += (x)
#define MSM_CHECK_COL_END(col) MSM_CHECK(chk_col_ == (unsigned __int128)(col))

#include "host_curve.hpp"
#include "te.hpp"

using namespace msm;

// ---- base-field primitives (curve ids 0/1 select the modulus) ------------------------------------------------
template <class F>
static void t_fe_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  Modulus<F> md;
  Fe x, y, z;
  uint32_t wa[12], wb[12], wo[12];
  memcpy(wa, a, 48);
  memcpy(wb, b, 48);
  fe_from_abi<F>(x, wa, md);
  fe_from_abi<F>(y, wb, md);
  fe_mul<F>(z, x, y, md);
  fe_to_abi<F>(wo, z, md);
  memcpy(out, wo, 48);
}

template <class F>
static void t_fe_sqr(const uint8_t* a, uint8_t* out) {
  Modulus<F> md;
  Fe x, z;
  uint32_t wa[12], wo[12];
  memcpy(wa, a, 48);
  fe_from_abi<F>(x, wa, md);
  fe_sqr<F>(z, x, md);
  fe_to_abi<F>(wo, z, md);
  memcpy(out, wo, 48);
}

// worst-case limbs for the column bounds: every limb at the largest value a multiply may see; only the checker matters
template <class F>
static void t_fe_extreme(int) {
  Modulus<F> md;
  Fe x, z;
  for (int i = 0; i < NL - 1; i++) x.v[i] = (1u << 30) - 1;
  x.v[NL - 1] = F::P[NL - 1] * 16;  // keeps the VALUE below 32p while every other limb sits at its bound
  fe_mul<F>(z, x, x, md);
  fe_sqr<F>(z, x, md);
  Fe y;
  for (int i = 0; i < NL - 1; i++) y.v[i] = (1u << 29) - 1;
  y.v[NL - 1] = F::P[NL - 1] * 8;
  fe_mul2<F>(z, y, y, y, y, md);
}

// k * a for small k through lazy limbs, then the weak reduction: out = canonical(k * a)
template <class F>
static void t_fe_weak_reduce(const uint8_t* a, int k, uint8_t* out) {
  Modulus<F> md;
  Fe x, z;
  uint32_t wa[12], wo[12];
  memcpy(wa, a, 48);
  fe_from_abi<F>(x, wa, md);
  fe_reduce<F>(x);
  // k = k1 * k2 (k1 <= 7, k2 <= 4) through lazy limbs: multiply, parallel carry, multiply again -> limbs < 2^31
  const int k1 = k > 7 ? 7 : k, k2 = k / k1;
  for (int i = 0; i < NL; i++) z.v[i] = x.v[i] * (uint32_t)k1;
  fe_carry(z);
  for (int i = 0; i < NL; i++) z.v[i] *= (uint32_t)k2;
  fe_weak_reduce<F>(z);
  Fe three_p;  // result must be < 3p
  for (int i = 0; i < NL; i++) three_p.v[i] = F::P[i] * 3;
  fe_normalize(three_p);
  MSM_CHECK(!fe_geq(z, three_p.v));
  fe_to_abi<F>(wo, z, md);
  memcpy(out, wo, 48);
}

template <class F>
static void t_fe_roundtrip(const uint8_t* a, uint8_t* out) {
  Modulus<F> md;
  Fe x;
  uint32_t wa[12], wo[12];
  memcpy(wa, a, 48);
  fe_from_abi<F>(x, wa, md);
  fe_to_abi<F>(wo, x, md);
  memcpy(out, wo, 48);
}

template <class F>
static void t_fe_inv(const uint8_t* a, uint8_t* out) {
  Modulus<F> md;
  Fe x, y;
  uint32_t wa[12], wo[12];
  memcpy(wa, a, 48);
  fe_from_abi<F>(x, wa, md);
  fe_inv<F>(y, x, md);
  fe_to_abi<F>(wo, y, md);
  memcpy(out, wo, 48);
}

// ---- coordinate-field and curve level (curve ids 0/1/2 = 377 G1, 381 G1, 377 G2) -----------------------------
template <class C>
static void t_el_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  using E = typename C::E;
  typename E::Md md;
  typename E::T x, y, z;
  uint32_t wa[E::WORDS], wb[E::WORDS], wo[E::WORDS];
  memcpy(wa, a, 4 * E::WORDS);
  memcpy(wb, b, 4 * E::WORDS);
  E::from_abi(x, wa, md);
  E::from_abi(y, wb, md);
  E::mul(z, x, y, md);
  E::to_abi(wo, z, md);
  memcpy(out, wo, 4 * E::WORDS);
}

template <class C>
static void t_el_inv(const uint8_t* a, uint8_t* out) {
  using E = typename C::E;
  typename E::Md md;
  typename E::T x, z;
  uint32_t wa[E::WORDS], wo[E::WORDS];
  memcpy(wa, a, 4 * E::WORDS);
  E::from_abi(x, wa, md);
  el_inv(z, x, md, (E*)nullptr);
  E::to_abi(wo, z, md);
  memcpy(out, wo, 4 * E::WORDS);
}

// acc = inf; for each i: acc += (+/-) points[i] (mixed add); out = normalised projective.
template <class C>
static void t_madd_chain(const uint8_t* pts, size_t stride, const uint8_t* neg, size_t n, uint8_t* out) {
  using E = typename C::E;
  typename E::Md md;
  XyzzT<typename E::T> acc;
  xyzz_set_inf<E>(acc);
  for (size_t i = 0; i < n; i++) {
    AffineT<typename E::T> p;
    if (affine_from_abi<E>(p, pts + i * stride, md)) continue;
    xyzz_madd<E>(acc, p, neg[i] != 0, false, md);
  }
  xyzz_to_projective_abi<E>(out, acc, md);
}

// out = (chain over first na points) + (chain over the remaining nb points), through the full XYZZ add.
template <class C>
static void t_add_chains(const uint8_t* pts, size_t stride, size_t na, size_t nb, uint8_t* out) {
  using E = typename C::E;
  typename E::Md md;
  XyzzT<typename E::T> a, b;
  xyzz_set_inf<E>(a);
  xyzz_set_inf<E>(b);
  for (size_t i = 0; i < na + nb; i++) {
    AffineT<typename E::T> p;
    if (affine_from_abi<E>(p, pts + i * stride, md)) continue;
    xyzz_madd<E>(i < na ? a : b, p, false, false, md);
  }
  xyzz_add<E>(a, b, md);
  xyzz_to_projective_abi<E>(out, a, md);
}

// sum k_i P_i by per-point double-and-add (XYZZ dbl + XYZZ add), then one running total.
template <class C>
static void t_msm_naive(const uint8_t* pts, size_t stride, const uint8_t* scalars, size_t n, uint8_t* out) {
  using E = typename C::E;
  typename E::Md md;
  XyzzT<typename E::T> total;
  xyzz_set_inf<E>(total);
  for (size_t i = 0; i < n; i++) {
    AffineT<typename E::T> p;
    if (affine_from_abi<E>(p, pts + i * stride, md)) continue;
    const uint8_t* k = scalars + 32 * i;
    XyzzT<typename E::T> r;
    xyzz_set_inf<E>(r);
    for (int bit = 255; bit >= 0; bit--) {
      if (!xyzz_is_inf<E>(r)) xyzz_dbl<E>(r, md);
      if ((k[bit >> 3] >> (bit & 7)) & 1) xyzz_madd<E>(r, p, false, false, md);
    }
    xyzz_add<E>(total, r, md);
  }
  xyzz_to_projective_abi<E>(out, total, md);
}

// projective (Jacobian, any Z) -> normalised projective, through XYZZ.
template <class C>
static void t_normalize(const uint8_t* in, uint8_t* out) {
  using E = typename C::E;
  typename E::Md md;
  XyzzT<typename E::T> a;
  xyzz_from_projective_abi<E>(a, in, md);
  xyzz_to_projective_abi<E>(out, a, md);
}

// ---- twisted-Edwards image of BLS12-377 G1 (te.hpp) ------------------------------------------------------------
using TF = Bls12_377_Fq;   // the shape the birational map runs in (14 x 28)
using TEF = TeFq;          // the shape the law runs in (13 x 29 unless built with -DMSM_TE_LIMBS29=0): te.hpp

// arkworks Affine image -> TE base record; false when the point has no image (or is flagged infinite).
static bool te_map_host(TeAffine& out, const uint8_t* img, const Modulus<TF>& md) {
  using E = FpEl<TF>;
  Affine p;
  if (affine_from_abi<E>(p, img, md)) return false;
  fe_reduce<TF>(p.x);
  fe_reduce<TF>(p.y);
  Fe u, v, w, den, inv;
  te_map_prepare<TF>(u, v, w, den, p, md);
  if (fe_is_zero_slow<TF>(den)) return false;
  fe_inv<TF>(inv, den, md);
  te_map_finish<TF>(out, u, v, w, inv, md);
  return true;
}

static int te_finish_host(uint8_t* out, const Xyzz& acc_te, const Modulus<TF>& md) {
  if (te_failed<TEF>(acc_te)) return 2;
  Xyzz sw, acc;
  te_point_to_28<TEF>(acc, acc_te, md);
  te_to_sw<TF>(sw, acc, md);
  xyzz_to_projective_abi<FpEl<TF>>(out, sw, md);
  return 0;
}

#define DISPATCH_F(curve, fn, ...)                        \
  switch (curve) {                                        \
    case 0: fn<Bls12_377_Fq>(__VA_ARGS__); return 0;      \
    case 1: fn<Bls12_381_Fq>(__VA_ARGS__); return 0;      \
    default: return -1;                                   \
  }
#define DISPATCH_C(curve, fn, ...)                        \
  switch (curve) {                                        \
    case 0: fn<Bls12_377_G1>(__VA_ARGS__); return 0;      \
    case 1: fn<Bls12_381_G1>(__VA_ARGS__); return 0;      \
    case 2: fn<Bls12_377_G2>(__VA_ARGS__); return 0;      \
    case 3: fn<Bls12_381_G2>(__VA_ARGS__); return 0;      \
    default: return -1;                                   \
  }

extern "C" {
long ht_check_failures(void) { return g_check_failures; }
const char* ht_first_failure(void) { return g_first_failure; }
void ht_reset_checks(void) { g_check_failures = 0; g_first_failure[0] = 0; }
int ht_fe_mul(int curve, const uint8_t* a, const uint8_t* b, uint8_t* out) { DISPATCH_F(curve, t_fe_mul, a, b, out) }
int ht_fe_sqr(int curve, const uint8_t* a, uint8_t* out) { DISPATCH_F(curve, t_fe_sqr, a, out) }
int ht_fe_extreme(int curve) { DISPATCH_F(curve, t_fe_extreme, 0) }
int ht_fe_weak_reduce(int curve, const uint8_t* a, int k, uint8_t* out) { DISPATCH_F(curve, t_fe_weak_reduce, a, k, out) }
int ht_fe_roundtrip(int curve, const uint8_t* a, uint8_t* out) { DISPATCH_F(curve, t_fe_roundtrip, a, out) }
int ht_fe_inv(int curve, const uint8_t* a, uint8_t* out) { DISPATCH_F(curve, t_fe_inv, a, out) }
int ht_el_mul(int curve, const uint8_t* a, const uint8_t* b, uint8_t* out) { DISPATCH_C(curve, t_el_mul, a, b, out) }
int ht_el_inv(int curve, const uint8_t* a, uint8_t* out) { DISPATCH_C(curve, t_el_inv, a, out) }
int ht_madd_chain(int curve, const uint8_t* pts, size_t stride, const uint8_t* neg, size_t n, uint8_t* out) { DISPATCH_C(curve, t_madd_chain, pts, stride, neg, n, out) }
int ht_add_chains(int curve, const uint8_t* pts, size_t stride, size_t na, size_t nb, uint8_t* out) { DISPATCH_C(curve, t_add_chains, pts, stride, na, nb, out) }
int ht_msm_naive(int curve, const uint8_t* pts, size_t stride, const uint8_t* scalars, size_t n, uint8_t* out) { DISPATCH_C(curve, t_msm_naive, pts, stride, scalars, n, out) }
int ht_normalize(int curve, const uint8_t* in, uint8_t* out) { DISPATCH_C(curve, t_normalize, in, out) }

// (Y - X, Y + X, 2dXY) -- the base record -- of the image of an arkworks Affine image, as three ABI Montgomery field images; 1 = no image.
int ht_te_map(const uint8_t* img, uint8_t* out) {
  Modulus<TF> md;
  TeAffine t;
  if (!te_map_host(t, img, md)) return 1;
  uint32_t w[36];
  fe_to_abi<TF>(w, t.ymx, md);
  fe_to_abi<TF>(w + 12, t.ypx, md);
  fe_to_abi<TF>(w + 24, t.td, md);
  memcpy(out, w, 144);
  return 0;
}
// sum of (+/-) points through te_madd, starting from the identity as the kernels do, mapped back: a Projective image.
// 1 = a point without image, 2 = an addition hit a vanishing denominator (Z = 0).
int ht_te_madd_chain(const uint8_t* pts, size_t stride, const uint8_t* neg, size_t n, uint8_t* out) {
  Modulus<TF> md;
  Modulus<TEF> tmd;
  Xyzz acc;
  te_set_identity<TEF>(acc);
  for (size_t i = 0; i < n; i++) {
    if (pts[i * stride + 96]) continue;
    TeAffine t28, t;
    if (!te_map_host(t28, pts + i * stride, md)) return 1;
    te_record_to<TEF>(t, t28, tmd);
    te_madd<TEF>(acc, t, neg[i] != 0, tmd);
    if (te_failed<TEF>(acc)) return 2;
  }
  return te_finish_host(out, acc, md);
}
// (chain over the first na points) + (chain over the rest) through the unified extended addition, then `dbl` doublings.
int ht_te_add_chains(const uint8_t* pts, size_t stride, size_t na, size_t nb, int dbl, uint8_t* out) {
  Modulus<TF> md;
  Modulus<TEF> tmd;
  Xyzz a, b;
  te_set_identity<TEF>(a);
  te_set_identity<TEF>(b);
  for (size_t i = 0; i < na + nb; i++) {
    if (pts[i * stride + 96]) continue;
    TeAffine t28, t;
    if (!te_map_host(t28, pts + i * stride, md)) return 1;
    te_record_to<TEF>(t, t28, tmd);
    te_madd<TEF>(i < na ? a : b, t, false, tmd);
  }
  te_add<TEF>(a, b, tmd);
  if (te_failed<TEF>(a)) return 2;
  for (int i = 0; i < dbl; i++) {
    te_dbl<TEF>(a, tmd);
    if (te_failed<TEF>(a)) return 2;
  }
  return te_finish_host(out, a, md);
}
// The 13 x 29 shape of BLS12-377 Fq on its own: out = a * b (ABI Montgomery images in and out), the product formed by
// fe_mul<Bls12_377_Fq29> between the two re-radixing steps (fe_28_to_29 / fe_29_to_28).
int ht_fe29_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  Modulus<TF> md;
  Modulus<Bls12_377_Fq29> md29;
  Fe x, y, x29, y29, z29, z;
  fe_from_abi<TF>(x, (const uint32_t*)a, md);
  fe_from_abi<TF>(y, (const uint32_t*)b, md);
  fe_reduce<TF>(x);
  fe_reduce<TF>(y);
  fe_28_to_29(x29, x, md29);
  fe_28_to_29(y29, y, md29);
  fe_mul<Bls12_377_Fq29>(z29, x29, y29, md29);
  fe_29_to_28(z, z29, md);
  uint32_t w[12];
  fe_to_abi<TF>(w, z, md);
  memcpy(out, w, 48);
  return 0;
}
// Worst-case limbs through the 13 x 29 Edwards law: every coordinate / record field is forced to the LARGEST limbs its class
// allows (class M: 2^29 - 1 everywhere and the top limb of 1.5p; records: canonical, i.e. the limbs of p - 1 raised to 2^29 - 1 below
// the top), `rounds` mixed additions and one full addition are run, and the MSM_CHECK column sums decide.  Values are meaningless.
int ht_te29_extreme(int rounds) {
  using F = Bls12_377_Fq29;
  Modulus<F> md;
  Fe big;
  for (int i = 0; i < F::N - 1; i++) big.v[i] = (1u << 29) - 1;
  big.v[F::N - 1] = F::P[F::N - 1] + (F::P[F::N - 1] >> 1) + 1;   // the top limb of 1.5p
  for (int i = F::N; i < NL; i++) big.v[i] = 0;
  Fe rec = big;
  rec.v[F::N - 1] = F::P[F::N - 1];
  Xyzz acc{big, big, big, big}, other{big, big, big, big};
  TeAffine b{rec, rec, rec};
  for (int r = 0; r < rounds; r++) {
    Xyzz t = acc;
    te_madd<F>(t, b, (r & 1) != 0, md);
    t = acc;
    te_madd<F, true>(t, b, (r & 1) != 0, md);
  }
  te_add<F>(acc, other, md);
  return 0;
}
}

// ---- the 64-bit host tail (host_fold64.hpp) against the generic arithmetic ------------------------------------------------
#include <type_traits>

#include "host_fold64.hpp"

template <class F>
static const Fp64& test_fp64() {
  static Fp64 f = [] {
    Fp64 g{};
    g.init<F>();
    if (F::P[0] == 1) {   // BLS12-377: the twisted-Edwards constants
      g.const_from_device<F>(g.two_d, Bls12_377_Te::K2D);
      g.const_from_device<F>(g.sqrt3, Bls12_377_Te::SQRT3);
      g.const_from_device<F>(g.fsc_sqrt3, Bls12_377_Te::FSC_SQRT3);
    }
    return g;
  }();
  return f;
}

// field: out = a * b (ABI Montgomery images in and out), through Fp64::mul / add / sub / invert; op selects
template <class F>
static int t_f64_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  const Fp64& f = test_fp64<F>();
  F64 x, y, z;
  memcpy(x.l, a, 48);
  memcpy(y.l, b, 48);
  switch (op) {
    case 0: f.mul(z, x, y); break;
    case 1: f.add(z, x, y); break;
    case 2: f.sub(z, x, y); break;
    case 3: f.invert(z, x); break;
    case 4: f.mul_portable(z, x, y); break;   // the C loop that runs where mulx / adcx / adox are missing
    default: return 1;
  }
  memcpy(out, z.l, 48);
  return 0;
}
extern "C" int ht_f64_op(int curve, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  return curve == 1 ? t_f64_op<Bls12_381_Fq>(op, a, b, out) : t_f64_op<Bls12_377_Fq>(op, a, b, out);
}

// Window sums S_w = k_w * P_w (affine images + 64-bit multipliers) folded with window size c, both ways; the two Projective
// images go to out_generic / out_fold64.  te = 1 (BLS12-377 only): the sums live on the twisted-Edwards image.
// Returns 0, or 2 when an Edwards addition hit a vanishing denominator (both implementations must agree on that too).
template <class E>
struct Fold64Of;   // the 64-bit field object that mirrors a device coordinate field
template <class F>
struct Fold64Of<FpEl<F>> {
  static const Fp64& get() { return test_fp64<F>(); }
};
template <class F, int NB>
struct Fold64Of<Fp2El<F, NB>> {
  static const Fp2_64<NB>& get() {
    static const Fp2_64<NB> f{&test_fp64<F>()};
    return f;
  }
};

template <class C>
static int t_fold_both(const uint8_t* pts, size_t stride, const uint64_t* mult, int windows, int c, int te, uint8_t* out_generic,
                       uint8_t* out_fold64) {
  using E = typename C::E;
  using F = typename E::Fld;
  using El = typename E::T;
  typename E::Md md;
  const auto& f = Fold64Of<E>::get();
  using FC = std::decay_t<decltype(f)>;
  constexpr size_t PB = 3 * 4 * E::WORDS;   // bytes of a Projective image
  XyzzT<El> sums[64];
  if (windows > 64) return 1;
  for (int w = 0; w < windows; w++) {
    AffineT<El> a;
    const bool inf = affine_from_abi<E>(a, pts + (size_t)w * stride, md);
    if (te) {
      if constexpr (std::is_same<E, FpEl<Bls12_377_Fq>>::value) {
        te_set_identity<F>(sums[w]);
        if (!inf) {
          TeAffine t;
          if (!te_map_host(t, pts + (size_t)w * stride, md)) return 1;
          for (int bit = 63; bit >= 0; bit--) {
            te_dbl<F>(sums[w], md);
            if ((mult[w] >> bit) & 1) te_madd<F>(sums[w], t, false, md);
          }
        }
      } else {
        return 1;
      }
    } else {
      xyzz_set_inf<E>(sums[w]);
      if (!inf) {
        const uint64_t k[4] = {mult[w], 0, 0, 0};
        xyzz_mul_u64x4<E>(sums[w], a, k, md);
      }
    }
  }
  XyzzT<El> g;
  XyzzG64<typename FC::El> h;
  bool ok_g = true, ok_h = true;
  if (te) {
    if constexpr (std::is_same<E, FpEl<Bls12_377_Fq>>::value) {
      ok_g = fold_windows_te<F>(g, sums, windows, c, md);
      ok_h = fold_windows_te64<F>(f, h, sums, windows, c);
    }
  } else {
    fold_windows<E>(g, sums, windows, c, md);
    fold_windows64<F>(f, h, sums, windows, c);
  }
  if (ok_g != ok_h) return 3;
  if (!ok_g) return 2;
  xyzz_to_projective_abi<E>(out_generic, g, md);
  sw64_to_abi(f, out_fold64, h);
  // the chunk sum as well: h + h against the generic doubling
  XyzzG64<typename FC::El> hh = h;
  sw64_add(f, hh, h);
  XyzzT<El> gg = g;
  xyzz_add<E>(gg, g, md);
  uint8_t b1[PB], b2[PB];
  xyzz_to_projective_abi<E>(b1, gg, md);
  sw64_to_abi(f, b2, hh);
  return memcmp(b1, b2, PB) == 0 ? 0 : 4;
}
extern "C" int ht_fold_both(int curve, const uint8_t* pts, size_t stride, const uint64_t* mult, int windows, int c, int te,
                            uint8_t* out_generic, uint8_t* out_fold64) {
  if (curve == 1) return te ? 1 : t_fold_both<Bls12_381_G1>(pts, stride, mult, windows, c, 0, out_generic, out_fold64);
  if (curve == 2) return te ? 1 : t_fold_both<Bls12_377_G2>(pts, stride, mult, windows, c, 0, out_generic, out_fold64);
  if (curve == 3) return te ? 1 : t_fold_both<Bls12_381_G2>(pts, stride, mult, windows, c, 0, out_generic, out_fold64);
  return t_fold_both<Bls12_377_G1>(pts, stride, mult, windows, c, te, out_generic, out_fold64);
}

// ---- the op table of devtest_ops.hpp on the host: the twin of libmsm_devtest.so's msm_devtest_run ------------------------
// Same raw limb records in, same limbs out (the device build replaces the portable multiply loops, the Montgomery step and
// the selects with GCN assembly); MSM_CHECK is armed here, so a record that would overflow a 64-bit column or underflow a
// biased subtraction is caught on this side.
#include "devtest_ops.hpp"

template <class C, bool TE>
static int t_devop(int op, const uint32_t* in, int iw, uint32_t* out, int ow, size_t n) {
  for (size_t i = 0; i < n; i++) {
    const uint32_t* a = in + i * iw;
    uint32_t* r = out + i * ow;
    switch (op) {
#define DT_CASE(OP) case OP: devtest_apply<C, OP>(a, r); break;
      DT_CASE(DT_FE_MUL)
      DT_CASE(DT_FE_SQR)
      DT_CASE(DT_FE_MUL2)
      DT_CASE(DT_NOT_AND_LMASK)
      DT_CASE(DT_FE_WEAK_REDUCE)
      DT_CASE(DT_EL_MUL)
      DT_CASE(DT_EL_SQR)
      DT_CASE(DT_EL_MUL_C)
      DT_CASE(DT_EL_MUL_C_BIG)
      DT_CASE(DT_EL_SQR_C)
      DT_CASE(DT_EL_MUL_SUB_C)
      DT_CASE(DT_MADD_COMMON)
      DT_CASE(DT_MADD)
      DT_CASE(DT_ADD)
      DT_CASE(DT_DBL)
      default:
        if constexpr (TE) {
          switch (op) {
            DT_CASE(DT_TE_MADD)
            DT_CASE(DT_TE_MADD_SWAPPED)
            DT_CASE(DT_TE_ADD)
            DT_CASE(DT_TE_DBL)
            default: return -1;
          }
        } else {
          return -1;
        }
#undef DT_CASE
    }
  }
  return 0;
}

// the op subset of the 13 x 29 shape (devtest_ops.hpp: DT_CURVE_TE29)
static int t_devop_te29(int op, const uint32_t* in, int iw, uint32_t* out, int ow, size_t n) {
  using C = Bls12_377_G1_29;
  for (size_t i = 0; i < n; i++) {
    const uint32_t* a = in + i * iw;
    uint32_t* r = out + i * ow;
    switch (op) {
#define DT_CASE(OP) case OP: devtest_apply<C, OP>(a, r); break;
      DT_CASE(DT_FE_MUL)
      DT_CASE(DT_TE_MADD)
      DT_CASE(DT_TE_MADD_SWAPPED)
      DT_CASE(DT_TE_ADD)
      DT_CASE(DT_TE_DBL)
#undef DT_CASE
      default: return -1;
    }
  }
  return 0;
}

extern "C" int ht_devop(int curve, int op, const uint32_t* in, uint32_t* out, size_t n) {
  int iw = 0, ow = 0;
  if (curve < 0 || curve > DT_CURVE_TE29 || !in || !out) return -1;
  if (curve == DT_CURVE_TE29) {
    if (!dt_op_in_te29(op)) return -1;
    devtest_shape(op, NL, iw, ow);
    return iw ? t_devop_te29(op, in, iw, out, ow, n) : -1;
  }
  devtest_shape(op, curve >= 2 ? 2 * NL : NL, iw, ow);
  if (!iw) return -1;
  switch (curve) {
    case 0: return t_devop<Bls12_377_G1, true>(op, in, iw, out, ow, n);
    case 1: return t_devop<Bls12_381_G1, false>(op, in, iw, out, ow, n);
    case 2: return t_devop<Bls12_377_G2, false>(op, in, iw, out, ow, n);
    default: return t_devop<Bls12_381_G2, false>(op, in, iw, out, ow, n);
  }
}

extern "C" int ht_devop_shape(int curve, int op, int* in_words, int* out_words) {
  if (!in_words || !out_words || curve < 0 || curve > DT_CURVE_TE29) return -1;
  if (curve == DT_CURVE_TE29) {
    *in_words = *out_words = 0;
    if (dt_op_in_te29(op)) devtest_shape(op, NL, *in_words, *out_words);
    return (*in_words) ? 0 : -1;
  }
  devtest_shape(op, curve >= 2 ? 2 * NL : NL, *in_words, *out_words);
  if (curve != 0 && ((op >= DT_TE_MADD && op <= DT_TE_DBL) || op == DT_TE_ADD_QUAD)) *in_words = *out_words = 0;
  return (*in_words) ? 0 : -1;
}
