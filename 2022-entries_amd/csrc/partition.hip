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

hipError_t PartLaunch::debug_finish(const PartPlan& p, const PartBuffers& b, const uint32_t* slot_keys, uint32_t nslots, hipStream_t st, char* what,
                                    size_t what_len, uint64_t* checks_done) {
  if (what && what_len) what[0] = 0;
#ifdef MSM_DEBUG
  uint32_t* const dbg = b.totals + 4;
  const uint32_t key_limit = p.bsets * p.half;
  hipLaunchKernelGGL(k_dbg_check_slots, dim3(part_ceil_div(nslots ? nslots : 1, 256)), dim3(256), 0, st, slot_keys, nslots, key_limit, dbg);
  hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  uint32_t h[4 + DBG_WORDS];
  e = hipMemcpy(h, b.totals, sizeof h, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return e;
  const uint32_t* v = h + 4;
  if (checks_done) *checks_done += v[DBG_CHECKS];
  static const char* const names[] = {"", "segment table has a gap or an overlap", "segment table does not end at the entry total",
                                      "an entry carries key bits its level has already resolved", "sorted keys decrease", "a key is out of range",
                                      "a value names a base outside this chunk", "a slot key after the accumulation is neither KEY_NONE nor a valid key"};
  if (v[DBG_DIGITS] != h[0])
    snprintf(what, what_len, "MSM_DEBUG: %u entries were grouped but the scalars hold %u non-zero digits", h[0], v[DBG_DIGITS]);
  else
    for (int i = DBG_SEG_GAP; i <= DBG_SLOT_KEY; i++)
      if (v[i]) {
        snprintf(what, what_len, "MSM_DEBUG: %s (%u violations; n = %u, c = %u, %u windows)", names[i], v[i], p.n, p.c, p.windows);
        break;
      }
#else
  (void)p; (void)b; (void)slot_keys; (void)nslots; (void)st; (void)what_len; (void)checks_done;
#endif
  return hipSuccess;
}

}  // namespace msm
