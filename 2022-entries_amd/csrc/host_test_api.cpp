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

using namespace msm;

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

// worst-case limbs for the column bounds: every limb 2^30 - 1 (the largest a multiply may see); only the checker matters
template <class F>
static void t_fe_extreme(int /*unused*/) {
  Modulus<F> md;
  Fe x, z;
  for (int i = 0; i < NL - 1; i++) x.v[i] = (1u << 30) - 1;
  x.v[NL - 1] = F::P[NL - 1] * 16;  // keeps the VALUE below 32p while every other limb sits at its bound
  fe_mul<F>(z, x, x, md);
  fe_sqr<F>(z, x, md);
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

// acc = inf; for each i: acc += (+/-) points[i] (mixed add); out = normalised projective.
template <class F>
static void t_madd_chain(const uint8_t* pts, size_t stride, const uint8_t* neg, size_t n, uint8_t* out144) {
  Modulus<F> md;
  Xyzz acc;
  xyzz_set_inf<F>(acc);
  for (size_t i = 0; i < n; i++) {
    Affine p;
    if (affine_from_abi<F>(p, pts + i * stride, md)) continue;
    xyzz_madd<F>(acc, p, neg[i] != 0, false, md);
  }
  xyzz_to_projective_abi<F>(out144, acc, md);
}

// out = (chain over first na points) + (chain over the remaining nb points), through the full XYZZ add.
template <class F>
static void t_add_chains(const uint8_t* pts, size_t stride, size_t na, size_t nb, uint8_t* out144) {
  Modulus<F> md;
  Xyzz a, b;
  xyzz_set_inf<F>(a);
  xyzz_set_inf<F>(b);
  for (size_t i = 0; i < na + nb; i++) {
    Affine p;
    if (affine_from_abi<F>(p, pts + i * stride, md)) continue;
    xyzz_madd<F>(i < na ? a : b, p, false, false, md);
  }
  xyzz_add<F>(a, b, md);
  xyzz_to_projective_abi<F>(out144, a, md);
}

// sum k_i P_i by per-point double-and-add (XYZZ dbl + XYZZ add), then one running total.
template <class F>
static void t_msm_naive(const uint8_t* pts, size_t stride, const uint8_t* scalars, size_t n, uint8_t* out144) {
  Modulus<F> md;
  Xyzz total;
  xyzz_set_inf<F>(total);
  for (size_t i = 0; i < n; i++) {
    Affine p;
    if (affine_from_abi<F>(p, pts + i * stride, md)) continue;
    const uint8_t* k = scalars + 32 * i;
    Xyzz r;
    xyzz_set_inf<F>(r);
    for (int bit = 255; bit >= 0; bit--) {
      if (!xyzz_is_inf<F>(r)) xyzz_dbl<F>(r, md);
      if ((k[bit >> 3] >> (bit & 7)) & 1) xyzz_madd<F>(r, p, false, false, md);
    }
    xyzz_add<F>(total, r, md);
  }
  xyzz_to_projective_abi<F>(out144, total, md);
}

// projective (Jacobian, any Z) -> normalised projective, through XYZZ.
template <class F>
static void t_normalize(const uint8_t* in144, uint8_t* out144) {
  Modulus<F> md;
  Xyzz a;
  xyzz_from_projective_abi<F>(a, in144, md);
  xyzz_to_projective_abi<F>(out144, a, md);
}

#define DISPATCH(curve, fn, ...)                          \
  switch (curve) {                                        \
    case 0: fn<Bls12_377_Fq>(__VA_ARGS__); return 0;      \
    case 1: fn<Bls12_381_Fq>(__VA_ARGS__); return 0;      \
    default: return -1;                                   \
  }

extern "C" {
long ht_check_failures(void) { return g_check_failures; }
const char* ht_first_failure(void) { return g_first_failure; }
void ht_reset_checks(void) { g_check_failures = 0; g_first_failure[0] = 0; }
int ht_fe_mul(int curve, const uint8_t* a, const uint8_t* b, uint8_t* out) { DISPATCH(curve, t_fe_mul, a, b, out) }
int ht_fe_sqr(int curve, const uint8_t* a, uint8_t* out) { DISPATCH(curve, t_fe_sqr, a, out) }
int ht_fe_extreme(int curve) { DISPATCH(curve, t_fe_extreme, 0) }
int ht_fe_roundtrip(int curve, const uint8_t* a, uint8_t* out) { DISPATCH(curve, t_fe_roundtrip, a, out) }
int ht_fe_inv(int curve, const uint8_t* a, uint8_t* out) { DISPATCH(curve, t_fe_inv, a, out) }
int ht_madd_chain(int curve, const uint8_t* pts, size_t stride, const uint8_t* neg, size_t n, uint8_t* out) { DISPATCH(curve, t_madd_chain, pts, stride, neg, n, out) }
int ht_add_chains(int curve, const uint8_t* pts, size_t stride, size_t na, size_t nb, uint8_t* out) { DISPATCH(curve, t_add_chains, pts, stride, na, nb, out) }
int ht_msm_naive(int curve, const uint8_t* pts, size_t stride, const uint8_t* scalars, size_t n, uint8_t* out) { DISPATCH(curve, t_msm_naive, pts, stride, scalars, n, out) }
int ht_normalize(int curve, const uint8_t* in, uint8_t* out) { DISPATCH(curve, t_normalize, in, out) }
}
