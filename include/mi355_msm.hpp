// mi355_msm.hpp -- header-only C++ mirror of the reference's Rust operator API over the C ABI (mi355_msm.h).
//
// The reference's host layer is Rust (P1A <entry>/src/lib.rs); this image has no Rust toolchain, so the host side above
// the C ABI is C++ where the reference is compiled code (the brief's rule).  Same names, argument meaning and error
// behaviour: `multi_scalar_mult_init(points) -> MultiScalarMultContext`, `multi_scalar_mult(ctx, points, scalars)
// -> Vec<G::Projective>` with batch_size = scalars.len() / points.len() (P1A 6block/src/lib.rs:54-109); a non-zero error
// code panics on the Rust side and throws here.  VariableBaseMSM::msm chops to the shorter slice
// (ARK ec/src/msm/variable_base/mod.rs:44-53).
#pragma once
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "mi355_msm.h"

namespace mi355 {

struct G1Affine {        // arkworks Affine image: size_of::<G1Affine>() == 104
  uint64_t x[6], y[6];
  uint8_t infinity;
  uint8_t pad[7];
};
struct BigInteger256 {
  uint64_t limbs[4];
};
struct G1Projective {    // arkworks Projective image (Jacobian), 144 bytes
  uint64_t x[6], y[6], z[6];
};
static_assert(sizeof(G1Affine) == 104 && sizeof(BigInteger256) == 32 && sizeof(G1Projective) == 144, "ABI layouts");

struct MsmError : std::runtime_error {
  int code;
  MsmError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

inline void check(RustError e) {
  if (e.code != 0) {
    std::string msg = e.message ? e.message : "(no message)";
    if (e.message) std::free(e.message);
    throw MsmError(e.code, "mi355_msm error " + std::to_string(e.code) + ": " + msg);
  }
}

struct MultiScalarMultContext {   // #[repr(C)] struct { context: *mut c_void }
  mi355_msm_ctx* context = nullptr;
  size_t npoints = 0;
  MultiScalarMultContext() = default;
  MultiScalarMultContext(const MultiScalarMultContext&) = delete;
  MultiScalarMultContext& operator=(const MultiScalarMultContext&) = delete;
  MultiScalarMultContext(MultiScalarMultContext&& o) noexcept : context(o.context), npoints(o.npoints) { o.context = nullptr; }
  ~MultiScalarMultContext() {
    if (context) {
      RustError e = mi355_msm_destroy(context);
      if (e.message) std::free(e.message);
    }
  }
};

inline MultiScalarMultContext multi_scalar_mult_init(const std::vector<G1Affine>& points, int curve = MI355_BLS12_377_G1) {
  MultiScalarMultContext ctx;
  check(mi355_msm_create(&ctx.context, curve, -1));
  check(mi355_msm_set_bases(ctx.context, points.data(), points.size(), sizeof(G1Affine)));
  ctx.npoints = points.size();
  return ctx;
}

inline std::vector<G1Projective> multi_scalar_mult(MultiScalarMultContext& ctx, const std::vector<G1Affine>& points,
                                                   const std::vector<BigInteger256>& scalars) {
  const size_t npoints = points.size();
  if (npoints != ctx.npoints) throw MsmError(-1, "multi_scalar_mult: context was initialised with a different point count");
  if (npoints == 0 || scalars.size() % npoints != 0) throw MsmError(-1, "multi_scalar_mult: scalars is not a whole number of batches");
  const size_t batch_size = scalars.size() / npoints;
  std::vector<G1Projective> ret(batch_size);
  check(mi355_msm_run(ctx.context, ret.data(), scalars.data(), npoints, batch_size));
  return ret;
}

// Stream-ordered run (mi355_msm_run_async; the role of ML bellman-cuda.h:48-75 msm_execute_async as P1A matter-labs/src/lib.rs:150-190
// uses it): `d_scalars` -- DEVICE memory, batch_size * npoints BigInteger256 -- must stay valid until wait() returns; the MSM is ordered
// after the work already enqueued in `stream` and runs on the context's own stream.  The job owns its output buffer.
struct MsmJob {
  mi355_msm_job* job = nullptr;
  std::vector<G1Projective> out;
  MsmJob() = default;
  MsmJob(const MsmJob&) = delete;
  MsmJob& operator=(const MsmJob&) = delete;
  MsmJob(MsmJob&& o) noexcept : job(o.job), out(std::move(o.out)) { o.job = nullptr; }
  bool done() const { return job == nullptr || mi355_msm_job_done(job) != 0; }
  std::vector<G1Projective>& wait() {
    if (job) {
      mi355_msm_job* j = job;
      job = nullptr;
      check(mi355_msm_job_wait(j));   // (releases the handle, whatever the status)
    }
    return out;
  }
  ~MsmJob() {
    if (job) {
      RustError e = mi355_msm_job_wait(job);
      if (e.message) std::free(e.message);
    }
  }
};

inline MsmJob multi_scalar_mult_async(MultiScalarMultContext& ctx, const void* d_scalars, size_t batch_size, void* stream = nullptr) {
  MsmJob j;
  j.out.resize(batch_size);
  check(mi355_msm_run_async(ctx.context, j.out.data(), d_scalars, ctx.npoints, batch_size, stream, nullptr, nullptr, &j.job));
  return j;
}

// VariableBaseMSM::msm_bigint shape: one stateless MSM, chopped to the shorter input.
inline G1Projective msm(const std::vector<G1Affine>& bases, const std::vector<BigInteger256>& scalars, int curve = MI355_BLS12_377_G1) {
  const size_t n = bases.size() < scalars.size() ? bases.size() : scalars.size();
  G1Projective out;
  check(mi355_msm(curve, &out, bases.data(), n, scalars.data(), sizeof(G1Affine)));
  return out;
}

// VariableBaseMSM::msm_chunks (ARK ec/src/msm/variable_base/mod.rs:165-199): `scalars` are Fr values (Montgomery form,
// converted on the device like into_bigint), must not outnumber the bases, pair up with the LAST scalars.size() bases, and
// are consumed `step` pairs at a time (the reference hard-codes 2^20); the partial sums are added.
inline G1Projective msm_chunks(const std::vector<G1Affine>& bases, const std::vector<BigInteger256>& scalars_fr,
                               size_t step = size_t(1) << 20, int curve = MI355_BLS12_377_G1) {
  if (scalars_fr.size() > bases.size() || step == 0) throw MsmError(-1, "msm_chunks: scalars_stream.len() <= bases_stream.len()");
  const size_t ns = scalars_fr.size(), skip = bases.size() - ns;
  MultiScalarMultContext ctx;
  check(mi355_msm_create(&ctx.context, curve, -1));
  check(mi355_msm_set_option(ctx.context, "scalars_montgomery", 1));
  std::vector<G1Projective> partials;
  for (size_t lo = 0; lo < ns; lo += step) {
    const size_t n = ns - lo < step ? ns - lo : step;
    check(mi355_msm_set_bases(ctx.context, bases.data() + skip + lo, n, sizeof(G1Affine)));
    partials.emplace_back();
    check(mi355_msm_run(ctx.context, &partials.back(), scalars_fr.data() + lo, n, 1));
  }
  G1Projective out;
  check(mi355_msm_fold(curve, &out, partials.data(), partials.size()));
  return out;
}

}  // namespace mi355
