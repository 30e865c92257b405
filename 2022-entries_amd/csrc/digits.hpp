// digits.hpp -- scalar windowing kernel (curve-independent apart from the scalar field constants).
// Reference behaviour: SPK msm/pippenger.cuh:116-123 (get_wval), CMB ProcessSignedDigits.cu:118-151 (signed digits).
#pragma once
#include "fp28.hpp"

namespace msm {

// Fr Montgomery form (a * 2^256 mod r, what arkworks' `Fr` holds) -> the plain integer a: one Montgomery reduction
// over 8 x 32-bit limbs.  This is `into_bigint` of VariableBaseMSM::msm (ARK ec/src/msm/variable_base/mod.rs:48-53,
// ff montgomery_backend.rs:445-465) and sppark's `mont` flag (SPK msm/pippenger.cuh:157-164).
template <class FR>
__device__ __forceinline__ void fr_from_montgomery(uint32_t (&s)[8]) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t m = s[0] * FR::RINV;
    uint64_t c = ((uint64_t)m * FR::R[0] + s[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      c += (uint64_t)m * FR::R[j] + s[j];
      s[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    s[7] = (uint32_t)c;
  }
  // result < 2r; bring it below r
  uint32_t t[8];
  int64_t b = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    b += (int64_t)s[j] - FR::R[j];
    t[j] = (uint32_t)b;
    b >>= 32;
  }
  if (b == 0) {
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = t[j];
  }
}

// One thread per scalar.  Digits d_w in [-2^(c-1), 2^(c-1)] with sum d_w 2^(cw) = k; windows*c >= 257 so
// the last carry always lands in a window.  Zero digits (and every digit of a base flagged infinite)
// get the sentinel key, which sorts behind every real bucket.
//
// Indices written into `vals` are absolute positions in the base table: idx0 + i (+ w * table_stride when the context
// holds precomputed tables 2^(c w) P, in which case all windows share one bucket set: key = |d| - 1, sentinel = 2^(c-1)).
template <class FR, bool MONT>
__global__ void __launch_bounds__(256) k_digits(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf,
                                                uint32_t n, uint32_t c, uint32_t windows, uint32_t idx0, uint32_t table_stride,
                                                uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint4* sp = reinterpret_cast<const uint4*>(scalars) + 2 * (size_t)i;
  uint4 lo = sp[0], hi = sp[1];
  uint32_t s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  if (MONT) fr_from_montgomery<FR>(s);
  const bool shared = table_stride != 0;
  const uint32_t half = 1u << (c - 1);
  const uint32_t sentinel = shared ? half : windows * half;
  const uint32_t wmask = (1u << c) - 1;
  uint32_t carry = 0;
  for (uint32_t w = 0; w < windows; w++) {
    uint32_t v = (s[0] & wmask) + carry;
#pragma unroll
    for (int j = 0; j < 7; j++) s[j] = (s[j] >> c) | (s[j + 1] << (32 - c));
    s[7] >>= c;
    const bool neg = v > half;
    const uint32_t d = neg ? (1u << c) - v : v;
    carry = neg ? 1u : 0u;
    const uint32_t idx = idx0 + i + w * table_stride;
    const bool dead = inf[idx] != 0;
    const uint32_t key = (d == 0 || dead) ? sentinel : (shared ? 0u : w * half) + (d - 1);
    keys[(size_t)w * n + i] = key;
    vals[(size_t)w * n + i] = idx | (neg ? 0x80000000u : 0u);
  }
}

}  // namespace msm
