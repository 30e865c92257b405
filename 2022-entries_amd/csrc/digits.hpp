// digits.hpp -- scalar-side helpers of the windowing (curve-independent apart from the scalar field constants); the digit
// extraction itself is fused into the first level of the bucket grouping (partition.hpp::next_digit).
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

}  // namespace msm
