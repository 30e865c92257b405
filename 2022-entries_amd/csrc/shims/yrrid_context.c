/* yrrid / combined-top-solutions FFI names over the MI355X engine (include/mi355_msm_shims.h).
 * int32 status: 0 = ok, otherwise the failing call's error code; the status is sticky like CMB MSM.cu:44,401-402.
 * The reference's preconditions (points % 65536 == 0, batches <= 16: CMB MSM.cu:364-372, 404-417) do not apply here. */
#define MI355_SHIM_YRRID
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
  /* The reference this shim stands in for negates every scalar with its top bit set (k' = r - k, -P: CMB ProcessSignedDigits.cu:
   * 10-20,123-128), i.e. it relies on bases of order r; so does this context unless MI355_MSM_ASSUME_SUBGROUP says otherwise. */
  const char* sub = getenv("MI355_MSM_ASSUME_SUBGROUP");
  if (y->ctx && !(sub && *sub)) take(y, mi355_msm_set_option(y->ctx, "assume_subgroup", 1));
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

/* ---- hex readers (CMB MSM.h:68-69) ------------------------------------------------------------------------------------
 * One token: skip white space, take hex digits up to the next white space or end of file; more than 2 * width digits or any
 * other character is an error.  The digits are most-significant first; the value is stored little-endian in `width` bytes. */
static int hex_value(int c) {
  if (c >= '0' && c <= '9') return c - '0';
  if (c >= 'a' && c <= 'f') return c - 'a' + 10;
  if (c >= 'A' && c <= 'F') return c - 'A' + 10;
  return -1;
}

static int is_space(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }

static int read_hex_token(FILE* f, uint8_t* out, int width) {
  char digits[2 * 96];
  int n = 0, c = getc(f);
  while (is_space(c)) c = getc(f);
  if (c == EOF) return 0;
  while (c != EOF && !is_space(c)) {
    if (hex_value(c) < 0 || n >= 2 * width) return 0;
    digits[n++] = (char)c;
    c = getc(f);
  }
  memset(out, 0, (size_t)width);
  for (int i = 0; i < n; i++) {   /* digit i from the END is nibble i of the little-endian image */
    const int v = hex_value(digits[n - 1 - i]);
    out[i / 2] |= (uint8_t)((i & 1) ? v << 4 : v);
  }
  return 1;
}

int32_t MSMReadHexPoints(uint8_t* pointsPtr, uint32_t count, const char* path) {
  FILE* f = (pointsPtr && path) ? fopen(path, "r") : NULL;
  if (!f) return -1;
  int32_t rc = 0;
  for (uint32_t i = 0; i < count && rc == 0; i++) {
    uint8_t* rec = pointsPtr + (size_t)i * 104;
    if (!read_hex_token(f, rec, 48) || !read_hex_token(f, rec + 48, 48)) rc = -1;
    memset(rec + 96, 0, 8);
  }
  fclose(f);
  return rc;
}

int32_t MSMReadHexScalars(uint8_t* scalarsPtr, uint32_t count, const char* path) {
  FILE* f = (scalarsPtr && path) ? fopen(path, "r") : NULL;
  if (!f) return -1;
  int32_t rc = 0;
  for (uint32_t i = 0; i < count && rc == 0; i++)
    if (!read_hex_token(f, scalarsPtr + (size_t)i * 32, 32)) rc = -1;
  fclose(f);
  return rc;
}
