/* BASELINE.json's literal entry point -- `msm(bases, scalars, n)` -- as its own shared object, one per curve
 * (libmi355msm_msm_{377,381}.so; SURVEY.md section 8b "symbol-name collisions": `msm` is too generic a name to live in the
 * main library).  The sppark convention for the rest: result through an out-pointer, RustError by value
 * (SPK poc/blst-cuda/cuda/pippenger_inf.cu:28-35, SPK util/rusterror.h:15-27); bases are arkworks G1Affine images
 * (104-byte stride), scalars BigInteger256.  Stateless: a pipelined upload + MSM per call (csrc/msm_stateless.hpp). */
#define MI355_SHIM_NORTH_STAR
#include "../../../include/mi355_msm_shims.h"

#if defined(FEATURE_BLS12_381)
#define SHIM_CURVE MI355_BLS12_381_G1
#else
#define SHIM_CURVE MI355_BLS12_377_G1
#endif

RustError msm(void* out, const void* bases, const void* scalars, size_t n) {
  return mi355_msm(SHIM_CURVE, out, bases, n, scalars, 104);
}
