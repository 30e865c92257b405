// C++ restatement of the reference's `msm_correctness` test (P1A combined-top-solutions/tests/msm.rs:15-40) against the
// harness-named FFI (include/mi355_msm_shims.h) and the C++ mirror of the Rust operator API (include/mi355_msm.hpp):
// generate points/scalars, run the accelerator for `batches` batches on one context, compare each batch with the CPU
// oracle (here: oracle/liboracle.so standing in for arkworks' VariableBaseMSM).  Built and run by tests/test_harness_cpp.py,
// once per curve: the reference selects the curve at compile time with a cargo feature that becomes -DFEATURE_BLS12_377 /
// -DFEATURE_BLS12_381 (P1A 6block/build.rs:9,82), and so does this file (it then links libmi355msm_zprize_381.so).
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <random>
#include <vector>

#define MI355_SHIM_ZPRIZE
#include "mi355_msm.hpp"
#include "mi355_msm_shims.h"

typedef int (*oracle_msm_t)(int, const void*, size_t, const void*, size_t, void*, int);

#if defined(FEATURE_BLS12_381)
static const int CURVE = MI355_BLS12_381_G1;
static const uint64_t R_TOP = 0x73eda753299d7d48ull;   // top limb of the BLS12-381 group order (ARKC bls12_381/src/fields/fr.rs:4)
#else
static const int CURVE = MI355_BLS12_377_G1;
static const uint64_t R_TOP = 0x12ab655e9a2ca556ull;   // ARKC bls12_377/src/fields/fr.rs:24
#endif

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s liboracle.so npow [batches]\n", argv[0]);
    return 2;
  }
  void* h = dlopen(argv[1], RTLD_NOW);
  if (!h) {
    fprintf(stderr, "dlopen oracle: %s\n", dlerror());
    return 2;
  }
  oracle_msm_t oracle_msm = (oracle_msm_t)dlsym(h, "oracle_msm");
  const size_t npoints = (size_t)1 << atoi(argv[2]);
  const size_t batches = argc > 3 ? atoi(argv[3]) : 4;

  std::vector<mi355::G1Affine> points(npoints);
  mi355::check(mi355_msm_generate_points(CURVE, 7, npoints < 2048 ? npoints : 2048, npoints, points.data(),
                                         sizeof(mi355::G1Affine)));
  points[3].infinity = 1;  // the baseline harness plants a point at infinity (P1A 6block/src/util.rs:26)
  std::mt19937_64 rng(12345);
  std::vector<mi355::BigInteger256> scalars(npoints * batches);
  for (auto& s : scalars) {
    for (int i = 0; i < 4; i++) s.limbs[i] = rng();
    s.limbs[3] %= R_TOP;  // below r
  }

  // (1) the Rust-API mirror
  mi355::MultiScalarMultContext ctx = mi355::multi_scalar_mult_init(points, CURVE);
  std::vector<mi355::G1Projective> got = mi355::multi_scalar_mult(ctx, points, scalars);
  if (got.size() != batches) return 1;

  // (2) the ZPrize harness FFI names
  RustContext rc{nullptr};
  mi355::check(mult_pippenger_init(&rc, points.data(), npoints, sizeof(mi355::G1Affine)));
  std::vector<mi355::G1Projective> got2(batches);
  mi355::check(mult_pippenger_inf(&rc, got2.data(), points.data(), npoints, batches, scalars.data(), sizeof(mi355::G1Affine)));

  int bad = 0;
  for (size_t b = 0; b < batches; b++) {
    mi355::G1Projective exp;
    if (oracle_msm(CURVE, points.data(), sizeof(mi355::G1Affine), scalars.data() + b * npoints, npoints, &exp, 0) != 0) return 2;
    if (memcmp(&exp, &got[b], sizeof exp) != 0 || memcmp(&exp, &got2[b], sizeof exp) != 0) {
      fprintf(stderr, "batch %zu differs from the CPU oracle\n", b);
      bad++;
    }
  }
  // (3) stateless msm() chops to the shorter input
  std::vector<mi355::BigInteger256> few(scalars.begin(), scalars.begin() + 10);
  mi355::G1Projective one = mi355::msm(points, few, CURVE), exp1;
  oracle_msm(CURVE, points.data(), sizeof(mi355::G1Affine), few.data(), 10, &exp1, 0);
  if (memcmp(&one, &exp1, sizeof one) != 0) bad++;
  // (4) the stream-ordered run (mi355::multi_scalar_mult_async): scalars in DEVICE memory, the call returns at once, wait() gives the same
  //     points.  This harness is plain g++ (no HIP headers), so the three runtime calls it needs come from the HIP library by name.
  {
    void* hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    typedef int (*malloc_t)(void**, size_t);
    typedef int (*memcpy_t)(void*, const void*, size_t, int);
    typedef int (*free_t)(void*);
    malloc_t hip_malloc = hip ? (malloc_t)dlsym(hip, "hipMalloc") : nullptr;
    memcpy_t hip_memcpy = hip ? (memcpy_t)dlsym(hip, "hipMemcpy") : nullptr;
    free_t hip_free = hip ? (free_t)dlsym(hip, "hipFree") : nullptr;
    if (!hip_malloc || !hip_memcpy || !hip_free) {
      fprintf(stderr, "libamdhip64.so: hipMalloc / hipMemcpy / hipFree not found\n");
      bad++;
    } else {
      void* d_scalars = nullptr;
      const size_t bytes = scalars.size() * sizeof(mi355::BigInteger256);
      if (hip_malloc(&d_scalars, bytes) != 0 || hip_memcpy(d_scalars, scalars.data(), bytes, 1 /* hipMemcpyHostToDevice */) != 0) {
        fprintf(stderr, "device allocation / copy failed\n");
        bad++;
      } else {
        mi355::MsmJob job = mi355::multi_scalar_mult_async(ctx, d_scalars, batches);
        std::vector<mi355::G1Projective>& got3 = job.wait();
        if (!job.done() || got3.size() != batches || memcmp(got3.data(), got.data(), batches * sizeof(mi355::G1Projective)) != 0) {
          fprintf(stderr, "the asynchronous run differs from the synchronous one\n");
          bad++;
        }
      }
      if (d_scalars) hip_free(d_scalars);
    }
  }
  printf("msm_correctness curve=%d npoints=2^%s batches=%zu: %s\n", CURVE, argv[2], batches, bad ? "FAILED" : "ok");
  return bad ? 1 : 0;
}
