/* yrrid / combined-top-solutions FFI names over the MI355X engine (include/mi355_msm_shims.h).
 * int32 status: 0 = ok, otherwise the failing call's error code; the status is sticky like CMB MSM.cu:44,401-402.
 * The reference's preconditions (points % 65536 == 0, batches <= 16: CMB MSM.cu:364-372, 404-417) do not apply here. */
#define MI355_SHIM_YRRID
#include <stdlib.h>

#include "../../../include/mi355_msm_shims.h"

typedef struct {
  mi355_msm_ctx* ctx;
  uint32_t points;
  int32_t error_state;
} yrrid_ctx;

static int32_t take(yrrid_ctx* y, RustError e) {
  if (e.code) {
    if (e.message) free(e.message);
    y->error_state = e.code;
  }
  return y->error_state;
}

void* MSMAllocContext(int32_t maxPoints, int32_t maxBatches) {
  (void)maxPoints; /* buffers are sized on demand */
  (void)maxBatches;
  yrrid_ctx* y = (yrrid_ctx*)calloc(1, sizeof *y);
  if (!y) return NULL;
  take(y, mi355_msm_create_env(&y->ctx, MI355_BLS12_377_G1));
  return y;
}

int32_t MSMFreeContext(void* context) {
  yrrid_ctx* y = (yrrid_ctx*)context;
  if (!y) return -1;
  int32_t rc = y->ctx ? take(y, mi355_msm_destroy(y->ctx)) : y->error_state;
  free(y);
  return rc;
}

int32_t MSMPreprocessPoints(void* context, void* affinePointsPtr, uint32_t points) {
  yrrid_ctx* y = (yrrid_ctx*)context;
  if (!y) return -1;
  if (y->error_state) return y->error_state;
  y->points = points;
  return take(y, mi355_msm_set_bases(y->ctx, affinePointsPtr, points, 104));
}

int32_t MSMRun(void* context, uint64_t* projectiveResultsPtr, void* scalarsPtr, uint32_t scalars) {
  yrrid_ctx* y = (yrrid_ctx*)context;
  if (!y) return -1;
  if (y->error_state) return y->error_state;
  if (y->points == 0 || scalars % y->points != 0) return y->error_state = -1;
  return take(y, mi355_msm_run(y->ctx, projectiveResultsPtr, scalarsPtr, y->points, scalars / y->points));
}
