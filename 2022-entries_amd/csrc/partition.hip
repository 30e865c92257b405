// partition.hip -- the bucket-grouping kernels (partition.hpp) instantiated for the two scalar fields, with and without the
// Fr-Montgomery entry (row f3).  Curve-independent otherwise; its own translation unit so that it compiles in parallel.
#include "launch.hpp"
#include "partition.hpp"

namespace msm {

int PartLaunch::run(int scalar_field, bool montgomery, const uint32_t* d_scalars, const uint8_t* d_inf, const PartPlan& p, const PartBuffers& b,
                    hipStream_t st, hipEvent_t mid, hipError_t& err) {
  if (scalar_field == 1)
    return montgomery ? part_run<Bls12_381_Fr, true>(d_scalars, d_inf, p, b, st, mid, err)
                      : part_run<Bls12_381_Fr, false>(d_scalars, d_inf, p, b, st, mid, err);
  return montgomery ? part_run<Bls12_377_Fr, true>(d_scalars, d_inf, p, b, st, mid, err)
                    : part_run<Bls12_377_Fr, false>(d_scalars, d_inf, p, b, st, mid, err);
}

}  // namespace msm
