// devtest_ops.hpp -- one table of arithmetic "ops" on RAW limb records, compiled twice from the same templates:
//   * by hipcc into libmsm_devtest.so (devtest.hip): one GPU thread per record -- the inline-GCN-assembly multiply chains,
//     v_bfi Montgomery step, v_cndmask selects and DPP quad permutes that only exist in the device build;
//   * by g++ into libmsm_hosttest.so (host_test_api.cpp): the portable loops with the limb-bound checker (MSM_CHECK) armed.
// tests/test_gpu_devtest.py feeds both the same records -- every limb at its documented bound, 0, p - 1, p, 2p - 1, random --
// and requires the same limbs back, then checks the values against Python big integers / the reference-pinned oracle.
// This is the element-wise parity SURVEY.md section 7 step 3 asks for (reference counterpart: SPK ff/mont_t.cuh:385-425,
// SPK ec/xyzz_t.hpp:178-249).  Test scaffolding: nothing in the engine links it.
#pragma once
#include "curve.hpp"
#include "te.hpp"

namespace msm {

enum DevtestOp : int {
  DT_FE_MUL = 0,        // in a, b (Fe)                    out Fe
  DT_FE_SQR = 1,        // in a                            out Fe
  DT_FE_MUL2 = 2,       // in a, b, c, d                   out Fe        a*b + c*d
  DT_NOT_AND_LMASK = 3, // in 1 word                       out 1 word
  DT_EL_MUL = 4,        // in a, b (T)                     out T         E::mul  (any lazy operands)
  DT_EL_SQR = 5,        // in a                            out T
  DT_EL_MUL_C = 6,      // in a, b (carried; b class M)    out T         E::mul_c<false>
  DT_EL_MUL_C_BIG = 7,  // in a, b (carried, b <= 18p)     out T         E::mul_c<true>
  DT_EL_SQR_C = 8,      // in a (carried)                  out T
  DT_EL_MUL_SUB_C = 9,  // in a, b, c, d                   out T         a*b - c*d as a stored coordinate
  DT_MADD_COMMON = 10,  // in acc (4 T), base (2 T), flags out acc (4 T), 1 word: returned "same x"
  DT_MADD = 11,         // in acc, base, flags             out acc, 0
  DT_ADD = 12,          // in acc, b (4 T each)            out acc
  DT_DBL = 13,          // in acc                          out acc
  DT_TE_MADD = 14,      // in acc (4 Fe), base (3 Fe), flags  out acc    BLS12-377 G1 only
  DT_TE_MADD_SWAPPED = 15,   // the k_accumulate_glds form: the caller swapped ymx / ypx of a negated base already
  DT_TE_ADD = 16,       // in acc, b                       out acc
  DT_TE_DBL = 17,       // in acc                          out acc
  DT_ADD_QUAD = 18,     // as DT_ADD, four lanes per record (device only)
  DT_TE_ADD_QUAD = 19,  // as DT_TE_ADD, four lanes per record (device only)
  DT_FE_WEAK_REDUCE = 20,    // in a (limbs < 2^31, value < 32p)  out Fe
  DT_COUNT = 21,
  DT_PAIR = 64          // + a G2 op in [DT_EL_MUL, DT_DBL]: the same records with two lanes per record (fp2pair.hpp; device only)
};

// words of one input / output record of `op` for coordinate elements of EW words (14 for Fp, 28 for Fp2); 0 = unknown op
MSM_HD void devtest_shape(int op, int EW, int& in_words, int& out_words) {
  in_words = out_words = 0;
  switch (op) {
    case DT_FE_MUL: in_words = 2 * NL; out_words = NL; break;
    case DT_FE_SQR: in_words = NL; out_words = NL; break;
    case DT_FE_MUL2: in_words = 4 * NL; out_words = NL; break;
    case DT_NOT_AND_LMASK: in_words = 1; out_words = 1; break;
    case DT_EL_MUL: case DT_EL_MUL_C: case DT_EL_MUL_C_BIG: in_words = 2 * EW; out_words = EW; break;
    case DT_EL_SQR: case DT_EL_SQR_C: in_words = EW; out_words = EW; break;
    case DT_EL_MUL_SUB_C: in_words = 4 * EW; out_words = EW; break;
    case DT_MADD_COMMON: case DT_MADD: in_words = 6 * EW + 1; out_words = 4 * EW + 1; break;
    case DT_ADD: case DT_ADD_QUAD: in_words = 8 * EW; out_words = 4 * EW; break;
    case DT_DBL: in_words = 4 * EW; out_words = 4 * EW; break;
    case DT_TE_MADD: case DT_TE_MADD_SWAPPED: in_words = 7 * NL + 1; out_words = 4 * NL; break;
    case DT_TE_ADD: case DT_TE_ADD_QUAD: in_words = 8 * NL; out_words = 4 * NL; break;
    case DT_TE_DBL: in_words = 4 * NL; out_words = 4 * NL; break;
    case DT_FE_WEAK_REDUCE: in_words = NL; out_words = NL; break;
    default: break;
  }
}

// "curve" id 4 of the test libraries: BLS12-377 Fq in its 13 x 29 limb shape (fp28.hpp) -- FE_MUL and the twisted-Edwards ops only.
// Records keep 14 words per field element (word 13 is 0), exactly as the kernels hold them.
struct Bls12_377_G1_29 {
  using E = FpEl<Bls12_377_Fq29>;
};
constexpr int DT_CURVE_TE29 = 4;
MSM_HD bool dt_op_in_te29(int op) { return op == DT_FE_MUL || (op >= DT_TE_MADD && op <= DT_TE_DBL) || op == DT_TE_ADD_QUAD; }

template <class T>
MSM_HD void dt_load(T& r, const uint32_t* w) {
  uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); i++) d[i] = w[i];
}
template <class T>
MSM_HD void dt_store(uint32_t* w, const T& r) {
  const uint32_t* s = reinterpret_cast<const uint32_t*>(&r);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); i++) w[i] = s[i];
}

// One record through one op; OP is a template parameter so that each op is its own (small) kernel.
template <class C, int OP>
MSM_HD void devtest_apply(const uint32_t* in, uint32_t* out) {
  using E = typename C::E;
  using F = typename E::Fld;
  using T = typename E::T;
  constexpr int EW = (int)(sizeof(T) / 4);
  typename E::Md md;
  if constexpr (OP == DT_FE_MUL) {
    Fe a, b, r;
    dt_load(a, in);
    dt_load(b, in + NL);
    fe_mul<F>(r, a, b, md);
    dt_store(out, r);
  } else if constexpr (OP == DT_FE_SQR) {
    Fe a, r;
    dt_load(a, in);
    fe_sqr<F>(r, a, md);
    dt_store(out, r);
  } else if constexpr (OP == DT_FE_MUL2) {
    Fe a, b, c, d, r;
    dt_load(a, in);
    dt_load(b, in + NL);
    dt_load(c, in + 2 * NL);
    dt_load(d, in + 3 * NL);
    fe_mul2<F>(r, a, b, c, d, md);
    dt_store(out, r);
  } else if constexpr (OP == DT_NOT_AND_LMASK) {
    out[0] = not_and_lmask(in[0]);
  } else if constexpr (OP == DT_FE_WEAK_REDUCE) {
    Fe a;
    dt_load(a, in);
    fe_weak_reduce<F>(a);
    dt_store(out, a);
  } else if constexpr (OP == DT_EL_MUL || OP == DT_EL_MUL_C || OP == DT_EL_MUL_C_BIG) {
    T a, b, r;
    dt_load(a, in);
    dt_load(b, in + EW);
    if constexpr (OP == DT_EL_MUL)
      E::mul(r, a, b, md);
    else if constexpr (OP == DT_EL_MUL_C)
      E::template mul_c<false>(r, a, b, md);
    else
      E::template mul_c<true>(r, a, b, md);
    dt_store(out, r);
  } else if constexpr (OP == DT_EL_SQR || OP == DT_EL_SQR_C) {
    T a, r;
    dt_load(a, in);
    if constexpr (OP == DT_EL_SQR) E::sqr(r, a, md); else E::sqr_c(r, a, md);
    dt_store(out, r);
  } else if constexpr (OP == DT_EL_MUL_SUB_C) {
    T a, b, c, d, r;
    dt_load(a, in);
    dt_load(b, in + EW);
    dt_load(c, in + 2 * EW);
    dt_load(d, in + 3 * EW);
    E::mul_sub_c(r, a, b, c, d, md);
    dt_store(out, r);
  } else if constexpr (OP == DT_MADD_COMMON || OP == DT_MADD) {
    XyzzT<T> acc;
    AffineT<T> base;
    dt_load(acc, in);
    dt_load(base, in + 4 * EW);
    const uint32_t flags = in[6 * EW];
    uint32_t ret = 0;
    if constexpr (OP == DT_MADD_COMMON)
      ret = xyzz_madd_common<E>(acc, base, (flags & 1) != 0, (flags & 2) != 0, md) ? 1u : 0u;
    else
      xyzz_madd<E>(acc, base, (flags & 1) != 0, (flags & 2) != 0, md);
    dt_store(out, acc);
    out[4 * EW] = ret;
  } else if constexpr (OP == DT_ADD) {
    XyzzT<T> acc, b;
    dt_load(acc, in);
    dt_load(b, in + 4 * EW);
    xyzz_add<E>(acc, b, md);
    dt_store(out, acc);
  } else if constexpr (OP == DT_DBL) {
    XyzzT<T> acc;
    dt_load(acc, in);
    xyzz_dbl<E>(acc, md);
    dt_store(out, acc);
  } else if constexpr (OP == DT_TE_MADD || OP == DT_TE_MADD_SWAPPED) {
    Xyzz acc;
    TeAffine base;
    dt_load(acc, in);
    dt_load(base, in + 4 * NL);
    const bool negate = (in[7 * NL] & 1) != 0;
    if constexpr (OP == DT_TE_MADD)
      te_madd<F, false>(acc, base, negate, md);
    else
      te_madd<F, true>(acc, base, negate, md);
    dt_store(out, acc);
  } else if constexpr (OP == DT_TE_ADD) {
    Xyzz acc, b;
    dt_load(acc, in);
    dt_load(b, in + 4 * NL);
    te_add<F>(acc, b, md);
    dt_store(out, acc);
  } else if constexpr (OP == DT_TE_DBL) {
    Xyzz acc;
    dt_load(acc, in);
    te_dbl<F>(acc, md);
    dt_store(out, acc);
  }
}

}  // namespace msm
