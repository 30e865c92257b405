// launch_pair_impl.hpp -- definitions of LaunchPair<E> (two lanes per G2 point: fp2pair.hpp, SwPairLaw); include only from
// kernels_<curve>p.hip.  Grids are twice those of Launch<E>: walking lane t is the pair of hardware lanes 2t, 2t + 1.
#pragma once
#include "launch.hpp"
#include "msm_kernels.hpp"

namespace msm {

template <class E>
struct PairLawOf;
template <class F, int NB>
struct PairLawOf<Fp2El<F, NB>> {
  using G = SwPairLaw<F, NB>;
};

inline uint32_t pair_blocks(uint64_t n) { return (uint32_t)((2 * n + 255) / 256); }

template <class E>
hipError_t LaunchPair<E>::accumulate(const uint2* entries, const uint32_t* n_real, uint32_t K, const AffineDevT<Fe2>* bases, SegOutT<Fe2> out,
                                     uint32_t nlanes, hipStream_t st) {
  using G = typename PairLawOf<E>::G;
  hipLaunchKernelGGL((k_accumulate_glds<G>), dim3(pair_blocks(nlanes)), dim3(256), 0, st, entries, n_real, K, bases, out, nlanes, (uint32_t*)nullptr);
  return hipGetLastError();
}

template <class E>
hipError_t LaunchPair<E>::segreduce(const XyzzDevT<Fe2>* in_slots, const uint32_t* in_keys, uint32_t n_in, uint32_t K, SegOutT<Fe2> out, uint32_t nlanes,
                                    hipStream_t st) {
  using G = typename PairLawOf<E>::G;
  hipLaunchKernelGGL((k_segreduce<G>), dim3(pair_blocks(nlanes)), dim3(256), 0, st, in_slots, in_keys, n_in, K, out, nlanes, (uint32_t*)nullptr);
  return hipGetLastError();
}

template <class E>
hipError_t LaunchPair<E>::bucket_reduce(bool first, const XyzzDevT<Fe2>* in_a, const XyzzDevT<Fe2>* in_x, uint32_t n_per_win, uint32_t L, uint32_t chunks,
                                        uint32_t windows, uint32_t out_stride, XyzzDevT<Fe2>* out_a, XyzzDevT<Fe2>* out_x, hipStream_t st) {
  using G = typename PairLawOf<E>::G;
  dim3 grid(pair_blocks((uint64_t)windows * chunks));
  if (first)
    hipLaunchKernelGGL((k_bucket_reduce<G, true>), grid, dim3(256), 0, st, in_a, in_x, n_per_win, L, chunks, windows, out_stride, out_a, out_x,
                       (uint32_t*)nullptr);
  else
    hipLaunchKernelGGL((k_bucket_reduce<G, false>), grid, dim3(256), 0, st, in_a, in_x, n_per_win, L, chunks, windows, out_stride, out_a, out_x,
                       (uint32_t*)nullptr);
  return hipGetLastError();
}

template <class E>
hipError_t LaunchPair<E>::reduce_scan_step(const XyzzDevT<Fe2>* in, const XyzzDevT<Fe2>* in2, XyzzDevT<Fe2>* out, uint32_t nb, uint32_t windows, uint32_t d,
                                           uint32_t mode, hipStream_t st) {
  using G = typename PairLawOf<E>::G;
  const uint64_t threads = (uint64_t)windows * (mode == 1 ? d : nb);
  hipLaunchKernelGGL((k_reduce_scan_step<G>), dim3(pair_blocks(threads)), dim3(256), 0, st, in, in2, out, nb, windows, d, mode, (uint32_t*)nullptr);
  return hipGetLastError();
}

template <class E>
hipError_t LaunchPair<E>::bucket_merge(XyzzDevT<Fe2>* total, const XyzzDevT<Fe2>* part, uint32_t n, hipStream_t st) {
  using G = typename PairLawOf<E>::G;
  hipLaunchKernelGGL((k_bucket_merge<G>), dim3(pair_blocks(n)), dim3(256), 0, st, total, part, n, (uint32_t*)nullptr);
  return hipGetLastError();
}

}  // namespace msm
