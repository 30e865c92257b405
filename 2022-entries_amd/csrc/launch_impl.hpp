// launch_impl.hpp -- definitions of Launch<E>; include only from kernels_<curve>.hip.
#pragma once
#include <type_traits>
#include "launch.hpp"
#include "msm_kernels.hpp"

namespace msm {

inline uint32_t launch_blocks(uint64_t n) { return (uint32_t)((n + 255) / 256); }

template <class E>
struct IsFp2 : std::false_type {};
template <class F, int NB>
struct IsFp2<Fp2El<F, NB>> : std::true_type {};

template <class E>
hipError_t Launch<E>::convert_bases(const uint8_t* in, size_t stride, uint32_t n, bool serialized, AffineDevT<El>* out, uint8_t* inf,
                                    hipStream_t st) {
  if (serialized)
    hipLaunchKernelGGL((k_convert_bases<E, true>), dim3(launch_blocks(n)), dim3(256), 0, st, in, stride, n, out, inf);
  else
    hipLaunchKernelGGL((k_convert_bases<E, false>), dim3(launch_blocks(n)), dim3(256), 0, st, in, stride, n, out, inf);
  return hipGetLastError();
}

template <class E>
hipError_t Launch<E>::accumulate(const uint2* entries, const uint32_t* n_real, uint32_t K,
                                 const AffineDevT<El>* bases, SegOutT<El> out, uint32_t nlanes, hipStream_t st, bool paired) {
  if constexpr (IsFp2<E>::value) {
    if (paired) return LaunchPair<E>::accumulate(entries, n_real, K, bases, out, nlanes, st);
  }
  // MSM_GATHER = 0 builds the one-lane-per-record walk instead (A/B: profiles/r02_ab_gather.txt)
#ifndef MSM_GATHER
#define MSM_GATHER 2
#endif
  if constexpr (MSM_GATHER == 2)
    hipLaunchKernelGGL((k_accumulate_glds<SwLaw<E>>), dim3(launch_blocks(nlanes)), dim3(256), 0, st, entries, n_real, K, bases, out, nlanes,
                       (uint32_t*)nullptr);
  else
    hipLaunchKernelGGL((k_accumulate<SwLaw<E>>), dim3(launch_blocks(nlanes)), dim3(256), 0, st, entries, n_real, K, bases, out, nlanes,
                       (uint32_t*)nullptr);
  return hipGetLastError();
}

template <class E>
hipError_t Launch<E>::sum_bases(const AffineDevT<El>* bases, const uint8_t* inf, uint32_t first, uint32_t n, uint32_t per_lane, SegOutT<El> out,
                                uint32_t nlanes, hipStream_t st) {
  hipLaunchKernelGGL((k_sum_bases<SwLaw<E>>), dim3(launch_blocks(nlanes)), dim3(256), 0, st, bases, inf, first, n, per_lane, out, nlanes, (uint32_t*)nullptr);
  return hipGetLastError();
}

template <class E>
hipError_t Launch<E>::segreduce(const XyzzDevT<El>* in_slots, const uint32_t* in_keys, uint32_t n_in, uint32_t K, SegOutT<El> out,
                                uint32_t nlanes, uint32_t quad_limit, hipStream_t st, bool paired) {
  if (nlanes <= quad_limit) {   // latency form: four lanes per addition (msm_kernels.hpp)
    hipLaunchKernelGGL((k_segreduce_quad<SwQuad<E>>), dim3(launch_blocks(4ull * nlanes)), dim3(256), 0, st, in_slots, in_keys, n_in, K, out, nlanes,
                       (uint32_t*)nullptr);
    return hipGetLastError();
  }
  if constexpr (IsFp2<E>::value) {
    if (paired) return LaunchPair<E>::segreduce(in_slots, in_keys, n_in, K, out, nlanes, st);
  }
  hipLaunchKernelGGL((k_segreduce<SwLaw<E>>), dim3(launch_blocks(nlanes)), dim3(256), 0, st, in_slots, in_keys, n_in, K, out, nlanes, (uint32_t*)nullptr);
  return hipGetLastError();
}

template <class E>
hipError_t Launch<E>::bucket_reduce(bool first, const XyzzDevT<El>* in_a, const XyzzDevT<El>* in_x, uint32_t n_per_win, uint32_t L,
                                    uint32_t chunks, uint32_t windows, uint32_t out_stride, XyzzDevT<El>* out_a, XyzzDevT<El>* out_x, hipStream_t st,
                                    bool paired) {
  if constexpr (IsFp2<E>::value) {
    if (paired) return LaunchPair<E>::bucket_reduce(first, in_a, in_x, n_per_win, L, chunks, windows, out_stride, out_a, out_x, st);
  }
  dim3 grid(launch_blocks((uint64_t)windows * chunks));
  if (first)
    hipLaunchKernelGGL((k_bucket_reduce<SwLaw<E>, true>), grid, dim3(256), 0, st, in_a, in_x, n_per_win, L, chunks, windows, out_stride, out_a, out_x,
                       (uint32_t*)nullptr);
  else
    hipLaunchKernelGGL((k_bucket_reduce<SwLaw<E>, false>), grid, dim3(256), 0, st, in_a, in_x, n_per_win, L, chunks, windows, out_stride, out_a, out_x,
                       (uint32_t*)nullptr);
  return hipGetLastError();
}

template <class E>
hipError_t Launch<E>::reduce_scan_step(const XyzzDevT<El>* in, const XyzzDevT<El>* in2, XyzzDevT<El>* out, uint32_t nb, uint32_t windows, uint32_t d, uint32_t mode,
                                       uint32_t quad_limit, hipStream_t st, bool paired) {
  const uint64_t threads = (uint64_t)windows * (mode == 1 ? d : nb);
  if (threads <= quad_limit) {
    hipLaunchKernelGGL((k_reduce_scan_step_quad<SwQuad<E>>), dim3(launch_blocks(4 * threads)), dim3(256), 0, st, in, in2, out, nb, windows, d, mode,
                       (uint32_t*)nullptr);
    return hipGetLastError();
  }
  if constexpr (IsFp2<E>::value) {
    if (paired) return LaunchPair<E>::reduce_scan_step(in, in2, out, nb, windows, d, mode, st);
  }
  hipLaunchKernelGGL((k_reduce_scan_step<SwLaw<E>>), dim3(launch_blocks(threads)), dim3(256), 0, st, in, in2, out, nb, windows, d, mode, (uint32_t*)nullptr);
  return hipGetLastError();
}

template <class E>
hipError_t Launch<E>::bucket_merge(XyzzDevT<El>* total, const XyzzDevT<El>* part, uint32_t n, hipStream_t st, bool paired) {
  if constexpr (IsFp2<E>::value) {
    if (paired) return LaunchPair<E>::bucket_merge(total, part, n, st);
  }
  hipLaunchKernelGGL((k_bucket_merge<SwLaw<E>>), dim3(launch_blocks(n)), dim3(256), 0, st, total, part, n, (uint32_t*)nullptr);
  return hipGetLastError();
}

template <class E>
hipError_t Launch<E>::pre_double(const AffineDevT<El>* in, const uint8_t* inf_in, uint32_t n, uint32_t c, XyzzDevT<El>* out, hipStream_t st) {
  hipLaunchKernelGGL((k_pre_double<E>), dim3(launch_blocks(n)), dim3(256), 0, st, in, inf_in, n, c, out);
  return hipGetLastError();
}

template <class E>
hipError_t Launch<E>::pre_normalize(const XyzzDevT<El>* in, uint32_t n, uint32_t J, El* prefix, AffineDevT<El>* out, uint8_t* inf_out,
                                    hipStream_t st) {
  hipLaunchKernelGGL((k_pre_normalize<E>), dim3(launch_blocks(((uint64_t)n + J - 1) / J)), dim3(256), 0, st, in, n, J, prefix, out, inf_out);
  return hipGetLastError();
}

}  // namespace msm
