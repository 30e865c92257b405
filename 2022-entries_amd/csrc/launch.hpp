// launch.hpp -- host-callable launchers of the per-curve kernels.  Declared here, defined in launch_impl.hpp and
// explicitly instantiated once per curve in kernels_<curve>.hip, so the three curve instantiations (each minutes of
// hipcc time: every field multiply is fully unrolled) compile in parallel and the engine TU stays small.
#pragma once
#include <hip/hip_runtime.h>

#include "host_curve.hpp"
#include "msm_types.hpp"
#include "partition_plan.hpp"
#include "te.hpp"

namespace msm {

template <class E>
struct Launch {
  using El = typename E::T;
  static hipError_t convert_bases(const uint8_t* in, size_t stride, uint32_t n, bool serialized, AffineDevT<El>* out, uint8_t* inf,
                                  hipStream_t st);
  // `paired` (G2 only, ignored over Fp; per context, option "g2_paired"): the throughput kernels run with every Fp2 value spread over
  // two neighbouring lanes (LaunchPair below, fp2pair.hpp) -- same records in memory, so the forms mix freely
  static hipError_t accumulate(const uint2* entries, const uint32_t* n_real, uint32_t K,
                               const AffineDevT<El>* bases, SegOutT<El> out, uint32_t nlanes, hipStream_t st, bool paired = false);
  // `quad_limit` (per context, option "quad_limit"): launches of at most that many additions use the four-lanes-per-addition
  // kernels (latency), larger ones one lane each (throughput)
  static hipError_t segreduce(const XyzzDevT<El>* in_slots, const uint32_t* in_keys, uint32_t n_in, uint32_t K, SegOutT<El> out,
                              uint32_t nlanes, uint32_t quad_limit, hipStream_t st, bool paired = false);
  static hipError_t pre_double(const AffineDevT<El>* in, const uint8_t* inf_in, uint32_t n, uint32_t c, XyzzDevT<El>* out, hipStream_t st);
  static hipError_t pre_normalize(const XyzzDevT<El>* in, uint32_t n, uint32_t J, El* prefix, AffineDevT<El>* out, uint8_t* inf_out,
                                  hipStream_t st);
  static hipError_t bucket_reduce(bool first, const XyzzDevT<El>* in_a, const XyzzDevT<El>* in_x, uint32_t n_per_win, uint32_t L,
                                  uint32_t chunks, uint32_t windows, uint32_t out_stride, XyzzDevT<El>* out_a, XyzzDevT<El>* out_x, hipStream_t st,
                                  bool paired = false);
  // small windows: one step of the scan-based reduction (k_reduce_scan_step)
  static hipError_t reduce_scan_step(const XyzzDevT<El>* in, const XyzzDevT<El>* in2, XyzzDevT<El>* out, uint32_t nb, uint32_t windows, uint32_t d, uint32_t mode,
                                     uint32_t quad_limit, hipStream_t st, bool paired = false);
  // carried buckets: total[b] += part[b] (k_bucket_merge)
  static hipError_t bucket_merge(XyzzDevT<El>* total, const XyzzDevT<El>* part, uint32_t n, hipStream_t st, bool paired = false);
  // anchored window: one fragment per lane of the plain sum of bases [first, first + n) (k_sum_bases)
  static hipError_t sum_bases(const AffineDevT<El>* bases, const uint8_t* inf, uint32_t first, uint32_t n, uint32_t per_lane, SegOutT<El> out,
                              uint32_t nlanes, hipStream_t st);
};

// The throughput kernels of a G2 curve with two lanes per point (SwPairLaw, laws.hpp); kernels_<curve>p.hip.  E = Fp2El<F, NB>.
template <class E>
struct LaunchPair {
  static hipError_t accumulate(const uint2* entries, const uint32_t* n_real, uint32_t K, const AffineDevT<Fe2>* bases, SegOutT<Fe2> out, uint32_t nlanes,
                               hipStream_t st);
  static hipError_t segreduce(const XyzzDevT<Fe2>* in_slots, const uint32_t* in_keys, uint32_t n_in, uint32_t K, SegOutT<Fe2> out, uint32_t nlanes,
                              hipStream_t st);
  static hipError_t bucket_reduce(bool first, const XyzzDevT<Fe2>* in_a, const XyzzDevT<Fe2>* in_x, uint32_t n_per_win, uint32_t L, uint32_t chunks,
                                  uint32_t windows, uint32_t out_stride, XyzzDevT<Fe2>* out_a, XyzzDevT<Fe2>* out_x, hipStream_t st);
  static hipError_t reduce_scan_step(const XyzzDevT<Fe2>* in, const XyzzDevT<Fe2>* in2, XyzzDevT<Fe2>* out, uint32_t nb, uint32_t windows, uint32_t d,
                                     uint32_t mode, hipStream_t st);
  static hipError_t bucket_merge(XyzzDevT<Fe2>* total, const XyzzDevT<Fe2>* part, uint32_t n, hipStream_t st);
};

// The twisted-Edwards fast path of BLS12-377 G1 (kernels_377te.hip).  `flags`: [0] += bases without an image (convert),
// [1] = 1 when an addition hit a vanishing denominator (any walking kernel).
struct LaunchTe {
  static constexpr uint32_t kDefaultQuadLimit = 1u << 18;   // tools/quad_limit_sweep.py: flat from 2^16 up, 2^18 best at 2^20 pairs
  static hipError_t convert(const AffineDev* in, const uint8_t* inf, uint32_t n, uint32_t J, Fe* prefix, TeAffineDev* out, uint32_t* flags,
                            hipStream_t st);
  static hipError_t accumulate(const uint2* entries, const uint32_t* n_real, uint32_t K,
                               const TeAffineDev* bases, SegOut out, uint32_t nlanes, uint32_t* flags, hipStream_t st);
  static hipError_t segreduce(const XyzzDev* in_slots, const uint32_t* in_keys, uint32_t n_in, uint32_t K, SegOut out, uint32_t nlanes,
                              uint32_t quad_limit, uint32_t* flags, hipStream_t st);
  static hipError_t bucket_reduce(bool first, const XyzzDev* in_a, const XyzzDev* in_x, uint32_t n_per_win, uint32_t L, uint32_t chunks,
                                  uint32_t windows, uint32_t out_stride, XyzzDev* out_a, XyzzDev* out_x, uint32_t* flags, hipStream_t st);
  static hipError_t reduce_scan_step(const XyzzDev* in, const XyzzDev* in2, XyzzDev* out, uint32_t nb, uint32_t windows, uint32_t d, uint32_t mode,
                                     uint32_t quad_limit, uint32_t* flags, hipStream_t st);
  static hipError_t sum_bases(const TeAffineDev* bases, const uint8_t* inf, uint32_t first, uint32_t n, uint32_t per_lane, SegOut out, uint32_t nlanes,
                              uint32_t* flags, hipStream_t st);
  // (no bucket_merge: the later chunks of a carried batch accumulate straight onto the stored buckets -- SegOutT::carry_in)
};

// Bucket grouping (partition.hip): digits + MSD partition of the (key, value) entries.  scalar_field: 0 = BLS12-377 Fr, 1 = BLS12-381 Fr
// (only the Montgomery conversion depends on it).  Returns the index of the entry buffer that holds the sorted entries.
struct PartLaunch {
  static int run(int scalar_field, bool montgomery, const uint32_t* d_scalars, const uint8_t* d_inf, const PartPlan& p, const PartBuffers& b,
                 hipStream_t st, hipEvent_t mid, hipError_t& err);
  // -DMSM_DEBUG builds only (otherwise a no-op that reports nothing): the slot-key check after the accumulation, then the stream is
  // synchronised and the violation counters of this chunk are read -- what[] names the first violated invariant, empty = all held.
  static hipError_t debug_finish(const PartPlan& p, const PartBuffers& b, const uint32_t* slot_keys, uint32_t nslots, hipStream_t st, char* what,
                                 size_t what_len, uint64_t* checks_done);
};

extern template struct Launch<Bls12_377_G1::E>;
extern template struct Launch<Bls12_381_G1::E>;
extern template struct Launch<Bls12_377_G2::E>;
extern template struct Launch<Bls12_381_G2::E>;
extern template struct LaunchPair<Bls12_377_G2::E>;
extern template struct LaunchPair<Bls12_381_G2::E>;

}  // namespace msm
