// kernels_377te.hip -- the walking kernels instantiated for the twisted-Edwards image of BLS12-377 G1 (te.hpp, laws.hpp),
// and the base converter.  Its own translation unit so that it compiles in parallel with the per-curve units.
#include "launch.hpp"
#include "msm_kernels.hpp"

namespace msm {

namespace {
using G = TeLaw<TeFq>;   // the limb shape of the Edwards path: te.hpp
inline uint32_t te_blocks(uint64_t n) { return (uint32_t)((n + 255) / 256); }
}  // namespace

hipError_t LaunchTe::convert(const AffineDev* in, const uint8_t* inf, uint32_t n, uint32_t J, Fe* prefix, TeAffineDev* out, uint32_t* flags,
                             hipStream_t st) {
  hipLaunchKernelGGL((k_te_convert<Bls12_377_Fq, TeFq>), dim3(te_blocks(((uint64_t)n + J - 1) / J)), dim3(256), 0, st, in, inf, n, J, prefix, out, flags);
  return hipGetLastError();
}

hipError_t LaunchTe::accumulate(const uint2* entries, const uint32_t* n_real, uint32_t K,
                                const TeAffineDev* bases, SegOut out, uint32_t nlanes, uint32_t* flags, hipStream_t st) {
  hipLaunchKernelGGL((k_accumulate_glds<G>), dim3(te_blocks(nlanes)), dim3(256), 0, st, entries, n_real, K, bases, out, nlanes, flags);
  return hipGetLastError();
}

hipError_t LaunchTe::sum_bases(const TeAffineDev* bases, const uint8_t* inf, uint32_t first, uint32_t n, uint32_t per_lane, SegOut out, uint32_t nlanes,
                               uint32_t* flags, hipStream_t st) {
  hipLaunchKernelGGL((k_sum_bases<G>), dim3(te_blocks(nlanes)), dim3(256), 0, st, bases, inf, first, n, per_lane, out, nlanes, flags);
  return hipGetLastError();
}

hipError_t LaunchTe::segreduce(const XyzzDev* in_slots, const uint32_t* in_keys, uint32_t n_in, uint32_t K, SegOut out, uint32_t nlanes,
                               uint32_t quad_limit, uint32_t* flags, hipStream_t st) {
  if (nlanes <= quad_limit)
    hipLaunchKernelGGL((k_segreduce_quad<TeQuad<TeFq>>), dim3(te_blocks(4ull * nlanes)), dim3(256), 0, st, in_slots, in_keys, n_in, K, out, nlanes, flags);
  else
    hipLaunchKernelGGL((k_segreduce<G>), dim3(te_blocks(nlanes)), dim3(256), 0, st, in_slots, in_keys, n_in, K, out, nlanes, flags);
  return hipGetLastError();
}

hipError_t LaunchTe::bucket_reduce(bool first, const XyzzDev* in_a, const XyzzDev* in_x, uint32_t n_per_win, uint32_t L, uint32_t chunks,
                                   uint32_t windows, uint32_t out_stride, XyzzDev* out_a, XyzzDev* out_x, uint32_t* flags, hipStream_t st) {
  dim3 grid(te_blocks((uint64_t)windows * chunks));
  if (first)
    hipLaunchKernelGGL((k_bucket_reduce<G, true>), grid, dim3(256), 0, st, in_a, in_x, n_per_win, L, chunks, windows, out_stride, out_a, out_x, flags);
  else
    hipLaunchKernelGGL((k_bucket_reduce<G, false>), grid, dim3(256), 0, st, in_a, in_x, n_per_win, L, chunks, windows, out_stride, out_a, out_x, flags);
  return hipGetLastError();
}

hipError_t LaunchTe::reduce_scan_step(const XyzzDev* in, const XyzzDev* in2, XyzzDev* out, uint32_t nb, uint32_t windows, uint32_t d, uint32_t mode,
                                      uint32_t quad_limit, uint32_t* flags, hipStream_t st) {
  const uint64_t threads = (uint64_t)windows * (mode == 1 ? d : nb);
  if (threads <= quad_limit)
    hipLaunchKernelGGL((k_reduce_scan_step_quad<TeQuad<TeFq>>), dim3(te_blocks(4 * threads)), dim3(256), 0, st, in, in2, out, nb, windows, d, mode, flags);
  else
    hipLaunchKernelGGL((k_reduce_scan_step<G>), dim3(te_blocks(threads)), dim3(256), 0, st, in, in2, out, nb, windows, d, mode, flags);
  return hipGetLastError();
}

}  // namespace msm
