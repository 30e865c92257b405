/* mi355_msm_shims.h -- the reference harnesses' OWN symbol names, as thin objects over include/mi355_msm.h.
 *
 * The reference tree uses the same symbol with different signatures in different entries:
 * `mult_pippenger_inf` is 5-argument/stateless in sppark's poc and 7-argument/context in the ZPrize harness, so each
 * flavour is its own small shared object, built per curve (the reference selects the curve at compile time with a
 * cargo feature -> -DFEATURE_BLS12_377 / -DFEATURE_BLS12_381, P1A 6block/build.rs:9,82):
 *
 *   libmi355msm_sppark_{377,381}.so   2022-entries_amd/csrc/shims/sppark_stateless.c
 *   libmi355msm_zprize_{377,381}.so   2022-entries_amd/csrc/shims/zprize_harness.c
 *   libmi355msm_yrrid_377.so          2022-entries_amd/csrc/shims/yrrid_context.c
 *   libmi355msm_msm_{377,381}.so      2022-entries_amd/csrc/shims/north_star_msm.c   (the literal `msm(bases, scalars, n)`)
 *
 * All of them link libmi355msm.so; none contains arithmetic.  The context-creating shims go through mi355_msm_create_env(), so
 * MI355_MSM_DEVICES=0,1,...,7 (or "all") turns an unchanged single-GPU harness into a sharded run over those MI355X.
 */
#ifndef MI355_MSM_SHIMS_H
#define MI355_MSM_SHIMS_H
#include "mi355_msm.h"

#ifdef __cplusplus
extern "C" {
#endif

#if defined(MI355_SHIM_SPPARK)
/* SPK poc/blst-cuda/cuda/pippenger_inf.cu:28-35 (Rust decl SPK poc/blst-cuda/src/lib.rs:29-35):
 * out = sum scalars[i] * points[i]; points are Affine images ffi_affine_sz bytes apart. */
RustError mult_pippenger_inf(void* out, const void* points, size_t npoints, const void* scalars, size_t ffi_affine_sz);
#endif

#if defined(MI355_SHIM_ZPRIZE)
/* P1A 6block/cuda/pippenger_inf.cu:44-53, 87-92 (Rust decl P1A 6block/src/lib.rs:23-40). */
typedef struct {
  void* context;
} RustContext;
RustError mult_pippenger_init(RustContext* context, const void* points, size_t npoints, size_t ffi_affine_sz);
RustError mult_pippenger_inf(RustContext* context, void* out, const void* points, size_t npoints, size_t batches,
                             const void* scalars, size_t ffi_affine_sz);
#endif

#if defined(MI355_SHIM_NORTH_STAR)
/* BASELINE.json north_star: "the sppark-style extern "C" msm(bases, scalars, n) FFI".  out receives one normalised
 * G1Projective image (144 B); bases are G1Affine images 104 bytes apart; scalars 32-byte integers.  Stateless. */
RustError msm(void* out, const void* bases, const void* scalars, size_t n);
#endif

#if defined(MI355_SHIM_YRRID)
/* CMB MSM.h:72-75 (Rust decl P1A combined-top-solutions/src/lib.rs:22-35): 0 on success, sticky non-zero otherwise. */
void* MSMAllocContext(int32_t maxPoints, int32_t maxBatches);
int32_t MSMFreeContext(void* context);
int32_t MSMPreprocessPoints(void* context, void* affinePointsPtr, uint32_t points);
int32_t MSMRun(void* context, uint64_t* projectiveResultsPtr, void* scalarsPtr, uint32_t scalars);
/* CMB MSM.h:68-69, MSM.cu:82-128 (behind SUPPORT_READING there; parseHex: prize4 yrrid C/Reader.c:10-54): whitespace-separated
 * hex tokens, most significant digit first, at most 2 x width digits, stored little-endian and zero-extended.  Points: x then y
 * (48 bytes each) into 104-byte G1Affine records whose flag word is cleared -- the values are taken as they are, i.e. the file
 * holds the in-memory (Montgomery) images MSMPreprocessPoints expects.  Scalars: 32 bytes each.  0 on success, -1 on a
 * missing file, a short file or a bad digit (the reference exits the process on a bad digit; a library should not). */
int32_t MSMReadHexPoints(uint8_t* pointsPtr, uint32_t count, const char* path);
int32_t MSMReadHexScalars(uint8_t* scalarsPtr, uint32_t count, const char* path);
#endif

#ifdef __cplusplus
}
#endif
#endif
