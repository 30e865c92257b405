/* The ZPrize prize1-msm harness FFI names (context + batches) over the MI355X engine (include/mi355_msm_shims.h).
 * Like the reference, the context is created by init and never freed by the harness (P1A 6block/cuda/pippenger_inf.cu:55). */
#define MI355_SHIM_ZPRIZE
#include <stdlib.h>
#include <string.h>

#include "../../../include/mi355_msm_shims.h"

#if defined(FEATURE_BLS12_381)
#define SHIM_CURVE MI355_BLS12_381_G1
#else
#define SHIM_CURVE MI355_BLS12_377_G1
#endif

static RustError shim_error(const char* msg) {
  RustError e;
  e.code = -1;
  e.message = strdup(msg);
  return e;
}

RustError mult_pippenger_init(RustContext* context, const void* points, size_t npoints, size_t ffi_affine_sz) {
  if (!context) return shim_error("mult_pippenger_init: null context");
  mi355_msm_ctx* ctx = NULL;
  RustError e = mi355_msm_create_env(&ctx, SHIM_CURVE);
  if (e.code) return e;
  e = mi355_msm_set_bases(ctx, points, npoints, ffi_affine_sz);
  if (e.code) {
    RustError d = mi355_msm_destroy(ctx);
    if (d.message) free(d.message);
    return e;
  }
  context->context = ctx;
  return e;
}

RustError mult_pippenger_inf(RustContext* context, void* out, const void* points, size_t npoints, size_t batches,
                             const void* scalars, size_t ffi_affine_sz) {
  (void)points; /* bases were uploaded by init, as in the reference */
  (void)ffi_affine_sz;
  if (!context || !context->context) return shim_error("mult_pippenger_inf: context not initialised");
  if (batches == 0) return shim_error("mult_pippenger_inf: batches must be > 0");
  return mi355_msm_run((mi355_msm_ctx*)context->context, out, scalars, npoints, batches);
}
