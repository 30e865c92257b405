// devtest.hip -- libmsm_devtest.so: the arithmetic ops of devtest_ops.hpp as gfx950 kernels, one thread (or one quad) per raw
// limb record.  TEST-ONLY library (tests/test_gpu_devtest.py): it is how the device-only code paths -- inline-assembly
// multiply chains, the v_bfi Montgomery step, v_cndmask_b32_e64 selects, DPP quad permutes -- are pinned element-wise at
// worst-case operands, which whole MSMs on random data never reach.  Nothing in libmi355msm.so links it.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "devtest_ops.hpp"
#include "fp2pair.hpp"
#include "host_curve.hpp"

namespace msm {

template <class C, int OP>
__global__ void __launch_bounds__(64) k_devtest(const uint32_t* __restrict__ in, int in_words, uint32_t* __restrict__ out, int out_words, uint32_t n) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  devtest_apply<C, OP>(in + (size_t)i * in_words, out + (size_t)i * out_words);
}

// four lanes per record: lane q owns coordinate q of both operands and of the result
template <class C, int OP>
__global__ void __launch_bounds__(64) k_devtest_quad(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
  using E = typename C::E;
  using T = typename E::T;
  constexpr int EW = (int)(sizeof(T) / 4);
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  const uint32_t i = t >> 2, q = t & 3;
  // (whole quads are in or out together: n records = 4n threads, blocks of 64)
  if (i >= n) return;
  typename E::Md md;
  T a, b;
  dt_load(a, in + (size_t)i * 8 * EW + q * EW);
  dt_load(b, in + (size_t)i * 8 * EW + 4 * EW + q * EW);
  if constexpr (OP == DT_ADD_QUAD)
    xyzz_add_quad<E>(a, b, q, md);
  else
    te_add_quad<typename E::Fld>(a, b, q, md);
  dt_store(out + (size_t)i * 4 * EW + q * EW, a);
}

// two lanes per record (fp2pair.hpp): lane h of a pair holds half h (c0 / c1) of every Fp2 value of the record.  The same op table
// as devtest_apply, instantiated over the paired coordinate policy; the limbs must equal the one-lane form's (and the host's).
template <class PE>
struct PairOf;
template <class F, int NB>
struct PairOf<Fp2El<F, NB>> {
  using E = Fp2PairEl<F, NB>;
};
template <class C, int OP>
__global__ void __launch_bounds__(64) k_devtest_pair(const uint32_t* __restrict__ in, int in_words, uint32_t* __restrict__ out, int out_words, uint32_t n) {
  using E = typename PairOf<typename C::E>::E;
  constexpr int EW = 2 * NL;
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  const uint32_t i = t >> 1, h = t & 1;
  if (i >= n) return;
  const uint32_t* src = in + (size_t)i * in_words + h * NL;    // half h of element e starts at src + e * EW
  uint32_t* dst = out + (size_t)i * out_words + h * NL;
  typename E::Md md;
  auto ld = [&](int e) { Fe r; dt_load(r, src + e * EW); return r; };
  auto st = [&](int e, const Fe& v) { dt_store(dst + e * EW, v); };
  if constexpr (OP == DT_EL_MUL || OP == DT_EL_MUL_C || OP == DT_EL_MUL_C_BIG) {
    Fe a = ld(0), b = ld(1), r;
    if constexpr (OP == DT_EL_MUL) E::mul(r, a, b, md);
    else if constexpr (OP == DT_EL_MUL_C) E::template mul_c<false>(r, a, b, md);
    else E::template mul_c<true>(r, a, b, md);
    st(0, r);
  } else if constexpr (OP == DT_EL_SQR || OP == DT_EL_SQR_C) {
    Fe a = ld(0), r;
    if constexpr (OP == DT_EL_SQR) E::sqr(r, a, md); else E::sqr_c(r, a, md);
    st(0, r);
  } else if constexpr (OP == DT_EL_MUL_SUB_C) {
    Fe a = ld(0), b = ld(1), c = ld(2), d = ld(3), r;
    E::mul_sub_c(r, a, b, c, d, md);
    st(0, r);
  } else if constexpr (OP == DT_MADD_COMMON || OP == DT_MADD) {
    XyzzT<Fe> acc{ld(0), ld(1), ld(2), ld(3)};
    AffineT<Fe> base{ld(4), ld(5)};
    const uint32_t flags = in[(size_t)i * in_words + 6 * EW];
    uint32_t ret = 0;
    if constexpr (OP == DT_MADD_COMMON)
      ret = xyzz_madd_common<E>(acc, base, (flags & 1) != 0, (flags & 2) != 0, md) ? 1u : 0u;
    else
      xyzz_madd<E>(acc, base, (flags & 1) != 0, (flags & 2) != 0, md);
    st(0, acc.x); st(1, acc.y); st(2, acc.zz); st(3, acc.zzz);
    if (h == 0) out[(size_t)i * out_words + 4 * EW] = ret;
  } else if constexpr (OP == DT_ADD) {
    XyzzT<Fe> acc{ld(0), ld(1), ld(2), ld(3)}, b{ld(4), ld(5), ld(6), ld(7)};
    xyzz_add<E>(acc, b, md);
    st(0, acc.x); st(1, acc.y); st(2, acc.zz); st(3, acc.zzz);
  } else if constexpr (OP == DT_DBL) {
    XyzzT<Fe> acc{ld(0), ld(1), ld(2), ld(3)};
    xyzz_dbl<E>(acc, md);
    st(0, acc.x); st(1, acc.y); st(2, acc.zz); st(3, acc.zzz);
  }
}

template <class C>
hipError_t launch_pair_op(int op, const uint32_t* d_in, int in_words, uint32_t* d_out, int out_words, uint32_t n) {
  switch (op) {
#define DT_PCASE(OP) case OP: hipLaunchKernelGGL((k_devtest_pair<C, OP>), dim3((2 * n + 63) / 64), dim3(64), 0, 0, d_in, in_words, d_out, out_words, n); return hipGetLastError();
    DT_PCASE(DT_EL_MUL)
    DT_PCASE(DT_EL_SQR)
    DT_PCASE(DT_EL_MUL_C)
    DT_PCASE(DT_EL_MUL_C_BIG)
    DT_PCASE(DT_EL_SQR_C)
    DT_PCASE(DT_EL_MUL_SUB_C)
    DT_PCASE(DT_MADD_COMMON)
    DT_PCASE(DT_MADD)
    DT_PCASE(DT_ADD)
    DT_PCASE(DT_DBL)
#undef DT_PCASE
    default: break;
  }
  return hipErrorInvalidValue;
}

template <class C, int OP>
hipError_t launch_one(const uint32_t* d_in, int in_words, uint32_t* d_out, int out_words, uint32_t n) {
  if constexpr (OP == DT_ADD_QUAD || OP == DT_TE_ADD_QUAD)
    hipLaunchKernelGGL((k_devtest_quad<C, OP>), dim3((4 * n + 63) / 64), dim3(64), 0, 0, d_in, d_out, n);
  else
    hipLaunchKernelGGL((k_devtest<C, OP>), dim3((n + 63) / 64), dim3(64), 0, 0, d_in, in_words, d_out, out_words, n);
  return hipGetLastError();
}

template <class C, bool TE>
hipError_t launch_op(int op, const uint32_t* d_in, int in_words, uint32_t* d_out, int out_words, uint32_t n) {
  switch (op) {
#define DT_CASE(OP) case OP: return launch_one<C, OP>(d_in, in_words, d_out, out_words, n);
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
    DT_CASE(DT_ADD_QUAD)
    default: break;
  }
  if constexpr (TE) {
    switch (op) {
      DT_CASE(DT_TE_MADD)
      DT_CASE(DT_TE_MADD_SWAPPED)
      DT_CASE(DT_TE_ADD)
      DT_CASE(DT_TE_DBL)
      DT_CASE(DT_TE_ADD_QUAD)
      default: break;
    }
  }
#undef DT_CASE
  return hipErrorInvalidValue;
}

// the op subset of the 13 x 29 shape (devtest_ops.hpp: DT_CURVE_TE29)
inline hipError_t launch_op_te29(int op, const uint32_t* d_in, int in_words, uint32_t* d_out, int out_words, uint32_t n) {
  using C = Bls12_377_G1_29;
  switch (op) {
#define DT_CASE(OP) case OP: return launch_one<C, OP>(d_in, in_words, d_out, out_words, n);
    DT_CASE(DT_FE_MUL)
    DT_CASE(DT_TE_MADD)
    DT_CASE(DT_TE_MADD_SWAPPED)
    DT_CASE(DT_TE_ADD)
    DT_CASE(DT_TE_DBL)
    DT_CASE(DT_TE_ADD_QUAD)
#undef DT_CASE
    default: break;
  }
  return hipErrorInvalidValue;
}

}  // namespace msm

extern "C" {

// words per input / output record of `op` on `curve` (0/0 for an op the curve does not have)
int msm_devtest_shape(int curve, int op, int* in_words, int* out_words) {
  if (!in_words || !out_words || curve < 0 || curve > msm::DT_CURVE_TE29) return -1;
  if (curve == msm::DT_CURVE_TE29) {
    *in_words = *out_words = 0;
    if (msm::dt_op_in_te29(op)) msm::devtest_shape(op, msm::NL, *in_words, *out_words);
    return (*in_words) ? 0 : -1;
  }
  if (op >= msm::DT_PAIR) {   // the two-lanes-per-record form of a G2 op: same records
    op -= msm::DT_PAIR;
    if (curve < 2 || op < msm::DT_EL_MUL || op > msm::DT_DBL) return -1;
  }
  msm::devtest_shape(op, curve >= 2 ? 2 * msm::NL : msm::NL, *in_words, *out_words);
  if (curve != 0 && op >= msm::DT_TE_MADD && op <= msm::DT_TE_DBL) *in_words = *out_words = 0;
  if (curve != 0 && op == msm::DT_TE_ADD_QUAD) *in_words = *out_words = 0;
  return (*in_words) ? 0 : -1;
}

// n records from HOST memory through op on the current device, results back to HOST memory.  0 = ok, otherwise a hipError_t
// (or -1 for a bad argument); a message goes to stderr.
int msm_devtest_run(int curve, int op, const uint32_t* in, uint32_t* out, size_t n) {
  int iw = 0, ow = 0;
  if (msm_devtest_shape(curve, op, &iw, &ow) != 0 || !in || !out || n == 0 || n > (1u << 24)) return -1;
  uint32_t *d_in = nullptr, *d_out = nullptr;
  hipError_t e = hipMalloc(&d_in, n * iw * 4);
  if (e == hipSuccess) e = hipMalloc(&d_out, n * ow * 4);
  if (e == hipSuccess) e = hipMemcpy(d_in, in, n * iw * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(d_out, 0xEE, n * ow * 4);
  if (e == hipSuccess && curve == msm::DT_CURVE_TE29) {
    e = msm::launch_op_te29(op, d_in, iw, d_out, ow, (uint32_t)n);
  } else if (e == hipSuccess && op >= msm::DT_PAIR) {
    e = curve == 2 ? msm::launch_pair_op<msm::Bls12_377_G2>(op - msm::DT_PAIR, d_in, iw, d_out, ow, (uint32_t)n)
                   : msm::launch_pair_op<msm::Bls12_381_G2>(op - msm::DT_PAIR, d_in, iw, d_out, ow, (uint32_t)n);
  } else if (e == hipSuccess) {
    switch (curve) {
      case 0: e = msm::launch_op<msm::Bls12_377_G1, true>(op, d_in, iw, d_out, ow, (uint32_t)n); break;
      case 1: e = msm::launch_op<msm::Bls12_381_G1, false>(op, d_in, iw, d_out, ow, (uint32_t)n); break;
      case 2: e = msm::launch_op<msm::Bls12_377_G2, false>(op, d_in, iw, d_out, ow, (uint32_t)n); break;
      default: e = msm::launch_op<msm::Bls12_381_G2, false>(op, d_in, iw, d_out, ow, (uint32_t)n); break;
    }
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, d_out, n * ow * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) fprintf(stderr, "msm_devtest_run(curve %d, op %d): %s\n", curve, op, hipGetErrorString(e));
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return (int)e;
}

}  // extern "C"
