/* sppark's stateless FFI name over the MI355X engine (see include/mi355_msm_shims.h). */
#define MI355_SHIM_SPPARK
#include "../../../include/mi355_msm_shims.h"

#if defined(FEATURE_BLS12_381)
#define SHIM_CURVE MI355_BLS12_381_G1
#else
#define SHIM_CURVE MI355_BLS12_377_G1
#endif

RustError mult_pippenger_inf(void* out, const void* points, size_t npoints, const void* scalars, size_t ffi_affine_sz) {
  return mi355_msm(SHIM_CURVE, out, points, npoints, scalars, ffi_affine_sz);
}
