// msm_engine.hip -- host orchestration of the MI355X MSM pipeline and the C ABI (include/mi355_msm.h).
//
// Role in the reference: the L1 "host orchestration" + L2 "C-ABI" layers of SURVEY.md section 1
// (SPK msm/pippenger.cuh:247-662 pippenger_t, CMB MSM.cu:149-532 MSMContext, ML msm.cu:97-468).
// One context = one device, one stream; bases are converted once and stay resident in HBM; a run is
//   digits + bucket grouping (partition.hpp) -> accumulate -> fragment merge -> bucket reduce -> host fold.
// Everything is enqueued on one stream with no host round-trip until the W window sums (W * 224 B) come back.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <strings.h>

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/mi355_msm.h"
#include "host_curve.hpp"
#include "host_fold64.hpp"
#include "host_pipeline.hpp"
#include "launch.hpp"

namespace {

using namespace msm;

// (the HIP-free host concurrency of host_pipeline.hpp throws PipelineError; a HIP failure IS one, so one handler serves both)
using HipFailure = msm_host::PipelineError;

#define HIP_OK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      char buf_[512];                                                                                  \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      throw HipFailure((int)e_, buf_);                                                                 \
    }                                                                                                  \
  } while (0)

[[noreturn]] void bad_arg(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw HipFailure(-1, buf);
}

RustError ok() { return RustError{0, nullptr}; }

// coordinates in Fq2 (two limb vectors per coordinate) / scalars modulo the BLS12-381 group order
inline bool is_g2(int curve) { return curve == MI355_BLS12_377_G2 || curve == MI355_BLS12_381_G2; }
inline bool is_381(int curve) { return curve == MI355_BLS12_381_G1 || curve == MI355_BLS12_381_G2; }

RustError fail(int code, const char* msg) {
  RustError e;
  e.code = code ? code : -1;
  e.message = strdup(msg ? msg : "unknown error");
  return e;
}

template <class Fn>
RustError guarded(Fn&& fn) {
  try {
    fn();
    return ok();
  } catch (const HipFailure& e) {
    return fail(e.code, e.what());
  } catch (const std::exception& e) {
    return fail(-1, e.what());
  } catch (...) {
    return fail(-1, "unknown C++ exception");
  }
}

// The calling thread's current HIP device is the CALLER's state (a torch process keeps allocating on it): every entry point
// that switches devices -- a context bound to another GPU, the shards of a sharded context -- restores it on the way out.
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() {
    if (hipGetDevice(&prev) != hipSuccess) {
      prev = -1;
      (void)hipGetLastError();
    }
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// `guarded` for the entry points that touch a device: the caller's current device is the same before and after.
template <class Fn>
RustError guarded_dev(Fn&& fn) {
  DeviceGuard keep;
  return guarded(fn);
}

// Device memory this library holds WITHOUT a caller knowing -- the cached contexts of the stateless call (msm_stateless.hpp) -- is
// given back before any allocation is allowed to fail: returns true when something was freed.
bool reclaim_idle_device_memory();

// MI355_MSM_GUARD_TAIL = 1 (diagnostic; tests/test_gpu_guard.py): every device buffer of this library is placed so that it ENDS where
// its mapping ends, with the address range behind it reserved but UNMAPPED (HIP virtual-memory API).  The hot kernels over-read by
// design -- the entry queue of k_accumulate_glds fetches whole sectors past a lane's last entry, its LDS-DMA gathers fetch record 0 for
// idle lanes -- and nothing but a slack convention (64 bytes behind the entry buffers) kept those reads inside an allocation; in this
// mode a read or write past a buffer's last (16-byte-rounded) byte is a GPU memory fault instead of a silent access to a neighbour.
// The reference keeps its only self-check disabled (CMB Partition4096.cu:419-432); device ASan cannot see the LDS-DMA intrinsic.
struct GuardedRange {
  void* va = nullptr;
  size_t va_bytes = 0, mapped = 0;
  hipMemGenericAllocationHandle_t handle{};
};
inline bool guard_tail_mode() {
  static const bool on = [] {
    const char* e = getenv("MI355_MSM_GUARD_TAIL");
    return e && *e && atol(e) != 0;
  }();
  return on;
}

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  GuardedRange* guard = nullptr;
  hipError_t guarded_alloc(size_t need) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran == 0) gran = (size_t)2 << 20;
    const size_t used = (need + 15) & ~(size_t)15;
    GuardedRange* g = new GuardedRange();
    g->mapped = (used + gran - 1) / gran * gran;
    g->va_bytes = g->mapped + gran;     // one granule behind the mapping stays unmapped: the guard
    e = hipMemAddressReserve(&g->va, g->va_bytes, gran, nullptr, 0);
    if (e == hipSuccess) {
      e = hipMemCreate(&g->handle, g->mapped, &prop, 0);
      if (e == hipSuccess) {
        e = hipMemMap(g->va, g->mapped, 0, g->handle, 0);
        if (e == hipSuccess) {
          hipMemAccessDesc acc{};
          acc.location = prop.location;
          acc.flags = hipMemAccessFlagsProtReadWrite;
          e = hipMemSetAccess(g->va, g->mapped, &acc, 1);
          if (e != hipSuccess) (void)hipMemUnmap(g->va, g->mapped);
        }
        if (e != hipSuccess) (void)hipMemRelease(g->handle);
      }
      if (e != hipSuccess) (void)hipMemAddressFree(g->va, g->va_bytes);
    }
    if (e != hipSuccess) {
      delete g;
      return e;
    }
    guard = g;
    p = (char*)g->va + (g->mapped - used);   // the buffer's last byte (rounded to 16) is the mapping's last byte
    return hipSuccess;
  }
  void reserve(size_t need) {
    if (need <= bytes) return;
    release();
    hipError_t e = guard_tail_mode() ? guarded_alloc(need) : hipMalloc(&p, need);
    if (e == hipErrorOutOfMemory) {
      (void)hipGetLastError();
      if (reclaim_idle_device_memory()) e = guard_tail_mode() ? guarded_alloc(need) : hipMalloc(&p, need);
    }
    if (e != hipSuccess) {
      p = nullptr;
      (void)hipGetLastError();   // clear the sticky per-thread error so that a retry with smaller buffers starts clean
      char buf[256];
      snprintf(buf, sizeof buf, "hipMalloc of %zu MiB failed: %s", need >> 20, hipGetErrorString(e));
      throw HipFailure((int)e, buf);
    }
    bytes = need;
  }
  void release() {
    if (guard) {
      // The address range stays RESERVED for the life of the process (47 bits of address space outlast any test): a later buffer never
      // lands on the addresses of a freed one.  Round 6: unmap + free + reserve + map of the SAME range within one process -- the small
      // buffers of the anchored window's sum of bases, replaced by the run's own a moment later -- made kernels of the second context
      // of a process fault ("write access to a read-only page") or return wrong sums under this mode and only under it: translations
      // of the old mapping were still in use.  With the range kept, a stale access faults instead of landing somewhere.
      (void)hipMemUnmap(guard->va, guard->mapped);
      (void)hipMemRelease(guard->handle);
      delete guard;
      guard = nullptr;
    } else if (p) {
      (void)hipFree(p);
    }
    p = nullptr;
    bytes = 0;
  }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

inline uint32_t ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
inline uint32_t ilog2_floor(uint64_t v) { uint32_t r = 0; while (v >>= 1) r++; return r; }

// ---- the anchored window (round 6) ------------------------------------------------------------------------------------------
// Signed digits carry: the window above the last FULL window of a canonical scalar holds `rem` = scalar_bits mod c bits plus the
// carry of the one below, so with rem <= 1 -- BLS12-377's Fr: 253 = 11 x 23, 252 = 12 x 21 = 14 x 18 -- it has (almost) no bits of its
// own and is non-zero all the same: for 57 % of the scalars at c = 21 (14 % would be the bit's own share), 14 % at c = 23.  The ZPrize
// winners remove those additions by halving the scalar (k -> r - k, -P: CMB ProcessSignedDigits.cu:10-20; option assume_subgroup here),
// which is only true when r P = O.
// Without any assumption: END the carry chain at the last full window -- its value v in [0, 2^c] is written 2^(c-1) + s,
// |s| <= 2^(c-1), the same buckets -- and add the constant part, 2^(c a + c - 1) x (the plain sum of the bases of the run), once on
// the host.  That sum depends on the bases only: computed by the pipeline itself (all scalars 1) the first time a context runs a
// given number of pairs and kept (AnchorSums).  A scalar of zero then costs ONE addition (its digit in the anchored window is
// -2^(c-1)) instead of none -- the price; contexts whose bases change with every call (the stateless pipeline) do not use it.
// Measured (profiles/r06_ab_anchor.txt): BLS12-377 G1 2^26, c = 21 anchored against c = 20 plain: -1.0 %; with tables (c = 23) 2^24: -2.5 %,
// G2 2^24: -1.2 % -- what signed-digit Pippenger has left once the limb shape is settled.
constexpr uint32_t kNoAnchor = 0xffffffffu;
// Windows' worth of additions per canonical, uniformly drawn scalar: `full` = scalar_bits / c whole windows and what lies above them.
// With q = 2^(bits-1) / r = 0.857 (BLS12-377), 0.552 (BLS12-381) -- the share of the scalars whose top bit is clear:
//   plain signed digits, rem = 0: the window above takes the carry of the top window, whose value stays below 2^c r / 2^bits: 1 - q;
//                        rem >= 1: it holds rem bits and a carry that comes half of the time: 1 - q 2^(1-rem) / 2
//   anchored: no carry arrives; the rem bits alone: 1 - q 2^(1-rem) (rem = 0: nothing is left above)
void eff_windows(int scalar_bits, int c, double& plain, double& anchored) {
  const int full = scalar_bits / c, rem = scalar_bits - full * c;
  const double q = scalar_bits == 253 ? 0.857 : 0.552;
  const double top_clear = rem == 0 ? 1.0 : std::min(1.0, q * std::ldexp(1.0, 1 - rem));   // P(the rem bits above the full windows are all zero)
  plain = full + (rem == 0 ? 1.0 - q : 1.0 - 0.5 * top_clear);
  anchored = full + (1.0 - top_clear);
}
// the window to anchor for window size c: the last full one, where that saves at least 1 % of the additions -- BLS12-377: c = 23 (1.3 %),
// 21 (3.4 %), 18 (2.9 %), 14, 12, ...; BLS12-381: c = 17 (2.9 %), 15.  (`always`: option anchor = 2 -- whatever the window size, for tests:
// the recoding is exact for any c)
uint32_t anchor_window(int c, int scalar_bits, bool always = false) {
  const int full = scalar_bits / c;
  if (full < 1) return kNoAnchor;
  double plain, anchored;
  eff_windows(scalar_bits, c, plain, anchored);
  return (always || (plain - anchored) >= 0.01 * plain) ? (uint32_t)(full - 1) : kNoAnchor;
}

// Window size.  Cost model in field-multiply units: one mixed add (10) per non-zero digit, ~50 per bucket for the
// bucket->window reduction (measured: 0.79 ns/bucket vs 0.15 ns/add).  Canonical scalars have `scalar_bits` bits, so the
// top window is only partly populated: with `rem` significant bits left it behaves like a full window, with none it
// only ever receives the signed-digit carry (about half of the scalars).  Buckets exist for all ceil(257/c) windows (any 256-bit
// scalar is legal), or for ONE window when precomputed tables let all digits share a bucket set.
// `levels` = precomputed table levels k (0: none; >= windows: one bucket set for all): windows g, g + G, ... share bucket set g,
// G = ceil(windows / k) sets exist.
// `anchor_min_c` (0: plain digits): window sizes from this one on are priced with their anchored window, where they have one.
int choose_window_bits(size_t n, int scalar_bits, bool shared_buckets, bool fold = false, int levels = 0, int anchor_min_c = 0) {
  // Between 2^12 and 2^19 pairs an MSM is latency, not throughput: what a window size costs is the launches it implies
  // (two scan steps of the bucket reduction per bit, a grouping pass more from 12 bits on, per-window steps of the level-1
  // tiles) and the model below does not see them.  Measured on all three curves (tools/small_c_sweep.py,
  // profiles/r02_small_c_sweep.txt): c = 8 up to ~2^15.5 and c = 11 up to 2^18 (2^17 for 255-bit scalars) are 4-14 % faster
  // than the model's choice (9..14).
  if (!shared_buckets && n > ((size_t)1 << 12)) {
    if (n < ((size_t)3 << 14)) return 8;
    if (n < ((size_t)3 << (scalar_bits > 253 ? 16 : 17))) return 11;
  }
  // Tables pay only where they buy a window: with the same number of windows as the table-free plan the additions are the same and
  // the gathers out of tens of GB of tables cost 3 % more (2^26: c = 21 with 2..6 levels 112 ms against 106 without tables; c = 23
  // with 6 levels 102.4 ms -- profiles/r04_table_levels_sweep.txt).  So a plan with tables only considers window sizes with FEWER
  // windows than the table-free choice.
  const int free_wins = shared_buckets ? (257 + choose_window_bits(n, scalar_bits, false, fold) - 1) / choose_window_bits(n, scalar_bits, false, fold) : 0;
  int best = 2;
  double best_cost = 1e300;
  for (int c = 2; c <= (shared_buckets ? 24 : 23); c++) {
    if (shared_buckets && c < 24 && (257 + c - 1) / c >= free_wins) continue;
    // c = 22 leaves 11 bits after level 1: a first generic pass over ONE bit (2 counters for 16384 entries of a tile) and segments
    // of 196 K entries for the second; it has the 12 windows of c = 23 and measures 10 % slower (same file)
    if (shared_buckets && c == 22) continue;
    // folded scalars (assume_subgroup) are < r/2: one bit less, and when c divides that the window above only takes the carry
    // of the scalars whose top digit exceeds 2^(c-1): (r/2 - 2^(bits-1)) / (r/2) = 14.5 % (BLS12-377), 44.8 % (BLS12-381)
    // (only from 2^25 pairs on: that is where it was measured to pay -- BLS12-377 G1 2^26, c = 21: 110.2 -> 108.3 ms; below, the model's
    //  margin is inside its error: G2 2^24 chose c = 18 and lost 4 %, profiles/r03_ab_fold.txt)
    const int bits = (fold && n >= ((size_t)1 << 25)) ? scalar_bits - 1 : scalar_bits;
    const int full = bits / c, rem = bits - full * c;
    double eff = full + (rem >= 2 ? 1.0 : (bits != scalar_bits && rem == 0 ? (scalar_bits == 253 ? 0.145 : 0.448) : 0.55));
    // (a window size chosen WITHOUT this term is still anchored where it can be: that is free.  Where the term may MOVE the choice is
    //  decided by measurement, Plan::plan: profiles/r06_ab_anchor.txt)
    if (anchor_min_c > 0 && !fold && c >= anchor_min_c && anchor_window(c, scalar_bits) != kNoAnchor) {
      double plain;
      eff_windows(scalar_bits, c, plain, eff);
    }
    const int wins = (257 + c - 1) / c;
    const double alloc = shared_buckets ? (double)(levels > 0 ? (wins + std::min(levels, wins) - 1) / std::min(levels, wins) : 1) : (double)wins;
    // from 22 bits on the grouping needs a second generic pass (level 1 resolves 10 bucket bits, a pass 10 more): +5.5 ms against 95 ms of
    // accumulation at 2^26, i.e. ~0.06 additions' worth per entry
    const double group = c >= 22 ? 0.06 : 0.0;
    const double cost = (eff + group * wins) * (double)n * 10.0 + alloc * (double)(1ull << (c - 1)) * 50.0;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

// The shape of a table set for `n` bases: window size c and the number of table levels actually built for `want_levels` asked for (0 = a
// level per window).  ONE place for build_tables, mi355_msm_plan and the automatic choice (the plan query derived its own).
struct TableShape {
  uint32_t c, levels, bsets;
};
TableShape table_shape(size_t n, int scalar_bits, long opt_window_bits, int want_levels) {
  TableShape t{};
  t.c = opt_window_bits ? (uint32_t)opt_window_bits : (uint32_t)choose_window_bits(n, scalar_bits, true, false, want_levels);
  const uint32_t all_windows = (257 + t.c - 1) / t.c;
  t.levels = want_levels > 0 ? std::min<uint32_t>((uint32_t)want_levels, all_windows) : all_windows;
  t.bsets = ceil_div(all_windows, t.levels);
  t.levels = ceil_div(all_windows, t.bsets);
  return t;
}

struct Plan {
  uint32_t c, windows, half, keybits;
  uint32_t bucket_windows;  // windows that own buckets: `windows`; with precomputed tables the bucket sets G = ceil(windows / levels)
  uint32_t levels;          // table levels in use (1 = none)
  uint32_t anchor;          // the anchored window (kNoAnchor: plain signed digits)
  uint64_t entries;     // windows * n
  uint32_t K, nlanes;   // accumulate geometry
  uint32_t segK;        // fragment-merge fan-in
  uint32_t logL0, logL; // bucket-reduce chunk sizes
  uint32_t T0;          // chunks per window on the first reduce level
  uint32_t L0;          // buckets per chunk there (2^logL0, or -- before a scan -- whatever fills the chip's SIMDs with one wave each)
  uint32_t scan_nb;     // elements per window the scan runs on: T0 rounded up to a power of two (the tail of a row stays empty)
  bool reduce_scan;     // finish the bucket reduction by a parallel scan (k_reduce_scan_step) ...
  bool scan_direct;     // ... directly on the buckets (small windows), or after one chunked level
};

}  // namespace

#ifndef MSM_G2_PAIRED_DEFAULT
#define MSM_G2_PAIRED_DEFAULT 31
#endif
constexpr long kDefaultG2Paired = MSM_G2_PAIRED_DEFAULT;

struct RcclState;   // the dlopen'ed RCCL entry points and one communicator per shard (sharded contexts only)

struct mi355_msm_ctx {
  int curve = 0;
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t copy_stream = nullptr;   // H2D of the next scalar batch while the current one computes
  struct AsyncWorker* async = nullptr; // mi355_msm_run_async: the context's worker thread and its job queue (created by the first job)
  hipEvent_t copy_ev[9] = {};   // batch parity 0/1 resident, pieces 0..6 of batch 0 resident
  size_t nbases = 0;
  DevBuf bases, inf;
  DevBuf stateless_raw[3];   // raw base records of the stateless pipeline (msm_stateless.hpp), kept with a cached context
  DevBuf scalars, entries[2], buckets, slots[2], slot_keys[2], red_a[2], red_x[2];
  DevBuf carry_buckets;      // the bucket array a multi-chunk batch carries from chunk to chunk (k_bucket_merge)
  std::vector<hipEvent_t> carry_ev;   // stage events of the chunks of such a batch (7 per chunk; they are read after the batch's only synchronisation)
  long opt_first_piece_div = 0;   // the first piece of a host-scalar batch is 1/div of it (0 = the default)
  long opt_carry = 1;        // 0: every chunk reduces its own buckets and the partial sums are added on the host (the A/B of profiles/r03_ab_carry.txt)
  DevBuf part_matrix, part_partial, part_segs[2], part_subjobs, part_counts, part_totals;   // bucket grouping scratch (partition_plan.hpp)
  void* pinned = nullptr;  // window sums land here
  size_t pinned_bytes = 0;
  hipEvent_t ev[8] = {};
  long opt_window_bits = 0, opt_lane_entries = 0, opt_max_chunk = 0, opt_seg_entries = 0, opt_scalars_montgomery = 0, opt_reduce_scan_log = 0;
  long opt_precompute = 0;
  long opt_table_levels = 0;      // with precompute: table levels k (0 = one per window); windows g, g + G, ... share bucket set g
  long opt_assume_subgroup = 0;   // 1: every base is in the order-r subgroup (r P = O), so a scalar k in (r/2, r) may run as (r - k)(-P)
  long opt_anchor = 1;            // option "anchor_window": 1 = end the signed-digit carry chain at the last full window where that saves additions
  bool anchor_armed = false;      // ... for the run under way: the sum of its bases is at hand (run_device_t)
  struct AnchorSum {
    size_t n;                     // bases [0, n)
    int shift;                    // -1: their plain sum; >= 0: 2^shift times it
    std::vector<uint8_t> pt;      // a HostTail<E>::Pt
  };
  std::vector<AnchorSum> anchor_sums;   // of the current base set (set_bases clears it); a handful of entries
  uint64_t anchor_sums_computed = 0;
  float anchor_sum_ms = 0;        // host wall time the most recent run spent computing such a sum (0: it was at hand)
  uint32_t last_anchor = kNoAnchor;   // the anchored window of the most recent chunk
  long opt_reduce_log_chunk = 0, opt_reduce_log_chunk0 = 0;
  long opt_reduce_scan = -1;      // 0: recursive chunked running sums only; otherwise the scan tail (default)
  long opt_reduce_fill = 0;       // waves per SIMD the first chunked level of the bucket reduction is cut for (0 = the default, 1)
  long opt_twisted_edwards = 1;   // BLS12-377 G1 only: accumulate on the twisted-Edwards image when every base has one
  // twisted-Edwards fast path (te.hpp): records for every table level; te_active is decided per base set
  DevBuf te_bases, flags;         // flags: u32[2] on the device, [0] bases without an image, [1] an addition failed
  uint32_t* h_flags = nullptr;    // pinned copy
  bool te_active = false;
  bool sw_level0_only = false;    // the short-Weierstrass tables were dropped after conversion (only level 0 is kept)
  uint64_t te_fallbacks = 0;      // runs repeated on the XYZZ path because an addition reported a vanishing denominator
  uint32_t te_fallback_streak = 0;   // consecutive chunks that fell back; two in a row demote the context to XYZZ for good
  uint64_t te_demotions = 0;
  uint64_t oom_backoffs = 0;      // chunks restarted with half the chunk size after a device allocation failed
  uint64_t peer_stagings = 0;     // sharded contexts: slices (bases or scalars) pulled from ANOTHER device's memory into a shard's own (msm_sharded.hpp)
  long opt_force_peer_staging = 0;   // test hook: take those branches although the source lies on the shard's own device (logical shards)
  uint32_t last_bucket_windows = 0, last_l1_bits = 0, last_l1_bins = 0, last_passes = 0;   // grouping geometry of the most recent chunk
  uint32_t auto_levels = 0;       // "precompute" = 2: the table levels chosen from the free HBM at the last set_bases (0 = none)
  uint64_t debug_checks = 0;      // -DMSM_DEBUG builds: invariant checks run so far (csrc/partition.hpp)
  size_t chunk_cap = 0;           // what the most recent run had to cap its chunks at after an allocation failed (0 = it never had to); reported, not kept
  size_t fitted_chunk = 0;        // largest chunk that has run with the current buffers and options (skips the fit query)
  bool fitted_tables = false;
  uint32_t fitted_c = 0;          // ... under this forced window size (0 = each chunk plans its own)
  long opt_mem_limit = 0;         // test hook: pretend the device has at most this many free bytes when sizing chunks
  long inject_alloc_failures = 0; // test hook: the next N work-buffer reservations of THIS context fail as if HBM were exhausted (-K: only the K-th from now)
  uint32_t quad_limit = LaunchTe::kDefaultQuadLimit;   // merge / scan launches of at most this many additions run four lanes per addition
  // G2 only (option "g2_paired"): which throughput kernels run with every Fp2 value spread over two lanes (fp2pair.hpp) --
  // bit 0 accumulate, 1 first level of the bucket reduction, 2 fragment merge, 3 scan steps, 4 bucket merge of carried batches
  long opt_g2_paired = kDefaultG2Paired;
  // sharded context (mi355_msm_create_sharded): this object then owns no device state itself, only the per-device children
  std::vector<mi355_msm_ctx*> shards;
  std::vector<size_t> shard_lo;   // bases [shard_lo[g], shard_lo[g+1]) live on shard g
  long opt_combine = 0;           // 0 auto (RCCL all-gather when available and the devices are distinct), 1 host fold only, 2 require RCCL
  RcclState* rccl = nullptr;
  uint64_t rccl_exchanges = 0;
  // precomputed tables (row f1): level w at bases[w * nbases ...] holds 2^(pre_c * w) * P; 0 = none
  uint32_t pre_c = 0, pre_windows = 0;
  bool bases_serialized = false;  // set only for the duration of mi355_msm_set_bases_serialized
  float last_ms[MI355_T_COUNT] = {};
  uint64_t last_info[8] = {};

  int scalar_bits() const { return (curve == MI355_BLS12_381_G1 || curve == MI355_BLS12_381_G2) ? 255 : 253; }
  // the anchored window is a throughput device: batches of 2^20 pairs and more, carried buckets (one window size per batch), no halved scalars
  // Window sizes the model may move to BECAUSE they have an anchored window (tools/calibrate_window_model.py with the option on,
  // profiles/r06_ab_anchor.txt).  BLS12-377: from c = 21 -- 2^26 pairs run at 21 instead of 20 (-1.0 %); below, c = 18 anchored loses
  // to its plain neighbours although it has fewer additions (2^23: 16.65 ms against 15.99 at c = 17; 2^24: 30.78 against 28.97 at c = 20).
  // BLS12-381 G1: any -- c = 17 (255 = 15 x 17) wins at 2^22 (11.26 ms against 12.06 at c = 16) and 2^23 (20.10 against 20.85 at c = 18).
  // BLS12-381 G2: not measured, left alone (no window size >= 21 has an anchored window there).
  int anchor_min_c() const { return curve == MI355_BLS12_381_G1 ? 2 : 21; }
  bool anchor_wanted(size_t n_batch) const {
    return opt_anchor != 0 && opt_carry != 0 && opt_assume_subgroup == 0 && n_batch >= (opt_anchor == 2 ? (size_t)1 : (size_t)1 << 20);
  }

  // use_tables = false plans the run WITHOUT the precomputed tables of this context (the XYZZ fallback of a
  // twisted-Edwards context, whose short-Weierstrass tables were dropped)
  // `force_c`: the window size of the whole batch when it runs as several chunks over carried buckets (run_device_t)
  Plan plan(size_t n, bool use_tables = true, uint32_t force_c = 0) const {
    Plan p{};
    const bool tables = pre_c && use_tables;
    if (tables)
      p.c = pre_c;
    else if (force_c)
      p.c = force_c;
    else
      p.c = (opt_window_bits && !pre_c) ? (uint32_t)opt_window_bits
                                        : (uint32_t)choose_window_bits(n, scalar_bits(), false, opt_assume_subgroup != 0, 0, anchor_wanted(n) ? anchor_min_c() : 0);
    p.anchor = anchor_armed ? anchor_window((int)p.c, scalar_bits(), opt_anchor == 2) : kNoAnchor;
    p.windows = (257 + p.c - 1) / p.c;
    p.levels = tables ? std::min<uint32_t>(pre_windows, p.windows) : 1;
    p.bucket_windows = ceil_div(p.windows, p.levels);
    p.levels = ceil_div(p.windows, p.bucket_windows);
    p.half = 1u << (p.c - 1);
    p.keybits = ilog2_floor(p.bucket_windows * p.half) + 1;   // bits of a bucket key (reported by mi355_msm_plan)
    p.entries = (uint64_t)p.windows * n;
    // entries per accumulate lane: 2^20 lanes up to 2^25 pairs (then 512 entries each); below that fewer, longer lanes win until the chip would go idle
    // (tools/small_k_sweep.py: 24 instead of 8 at 2^17..2^19 pairs: -4..-11 % wall; 36..64 instead of 18..36 at 2^20..2^21: -3 %)
    // (and 4 below 2^18 entries -- a few thousand pairs -- where even 8 additions in a row are a visible share: -4 %)
    const uint64_t k_auto = std::max<uint64_t>({p.entries >= (3u << 20) ? 24u : (p.entries < (1u << 18) ? 4u : 8u), p.entries >> 20, std::min<uint64_t>(64, p.entries >> 19)});
    // (capped at 512 since round 3, 256 before: the accumulation takes the same time for 128..1024 entries per lane, the fragment
    //  merge halves with the lane count -- 2^26: 109.0 -> 108.5 ms, 2^25: 57.0 -> 56.8, BLS12-381 flat; profiles/r03_ab_lane_entries.txt)
    uint32_t K = opt_lane_entries ? (uint32_t)opt_lane_entries : (uint32_t)std::min<uint64_t>(512, k_auto);
    // Block generations (round 6, tools/rounds_probe.py, profiles/r06_ab_lane_groups.txt).  The dispatcher hands the 256 CUs one block each
    // and the blocks of such a group run in step, so the accumulation takes ceil(working blocks / 256) x (one group's time): with the
    // pair count varied at K = 512 the BLS12-377 launch steps by 3.4 ms every 256 blocks and is flat in between (22 groups 76.0 ms, 23
    // 79.3-79.5, 24 82.7-82.8, 25 85.9).  2^26 pairs at c = 20 are 26.00 groups by luck (13 n / (512 x 256 lanes) / 256); the anchored
    // c = 21 is 24.29 -- a 25th generation for a third of a group.  So: the K nearest to the rule's whose expected working blocks fill their
    // last group to >= 98.5 % (here 520: 23.91 groups; 102.92 -> 101.27 ms same-box).  Working blocks = expected non-zero digits of uniform
    // canonical scalars / lanes of a block (128 where two hardware lanes share a point).  No safety margin on purpose: the power-of-two sizes
    // sit EXACTLY on a whole number of groups (13 x 2^26 / 2^25 = 26), the real count can only be below that capacity, and moving them
    // off it was measured to lose (BLS12-381 2^26: 132.5 -> 134.5 ms at K = 520).
    if (!opt_lane_entries && !opt_assume_subgroup && K >= 64) {
      K = (K + 7) & ~7u;   // (the candidates are the values a plan can have: multiples of 8, see below)
      double plain, anchored;
      eff_windows(scalar_bits(), (int)p.c, plain, anchored);
      // (a plan made outside a run -- mi355_msm_plan, the chunk fit -- expects the anchored window wherever a run would use it)
      const bool anchor_expected = (anchor_armed || anchor_wanted(n)) && anchor_window((int)p.c, scalar_bits(), opt_anchor == 2) != kNoAnchor;
      const double adds = (anchor_expected ? anchored : plain) * (double)n;
      const double per_group = ((curve == MI355_BLS12_377_G2 || curve == MI355_BLS12_381_G2) && (opt_g2_paired & 1) ? 128.0 : 256.0) * 256.0;
      auto fill = [&](long k) {
        const double g = adds / ((double)k * per_group);
        return g / std::ceil(g);
      };
      if (adds / ((double)K * per_group) >= 6.0) {
        long best = (long)K;
        double best_fill = fill(best);
        for (long step = 8; step <= 96 && best_fill < 0.985; step += 8)
          for (long k : {(long)K - step, (long)K + step})
            if (k >= 64 && fill(k) > best_fill + 0.004) {
              best = k;
              best_fill = fill(k);
            }
        K = (uint32_t)best;
      }
    }
    p.K = K >= 8 ? (K + 7) & ~7u : (K + 3) & ~3u;   // a lane's entries start on a 64-byte boundary (k_accumulate_glds refills its entry queue by whole sectors)
    p.nlanes = ceil_div(p.entries, p.K);
    p.segK = opt_seg_entries >= 4 ? (uint32_t)opt_seg_entries : (p.entries < (2u << 20) ? 4 : 8);   // small inputs: shallower levels (-1..-3 %)
    uint64_t nb = (uint64_t)p.bucket_windows * p.half;
    uint32_t l0 = nb > (1u << 18) ? ilog2_floor(nb >> 18) : 0;
    // chunks of 4 on the later (small, latency-bound) levels and on small inputs: -7 % wall at 2^14..2^18 against chunks of 8
    p.logL0 = std::min<uint32_t>(7, std::max<uint32_t>(2, l0));
    p.logL = 2;
    if (opt_reduce_log_chunk) p.logL0 = p.logL = (uint32_t)opt_reduce_log_chunk;
    if (opt_reduce_log_chunk0) p.logL0 = (uint32_t)opt_reduce_log_chunk0;   // first level only
    p.logL0 = std::min<uint32_t>(p.logL0, p.c - 1 ? p.c - 1 : 1);
    // The tail of the reduction is latency, not work: once a window is down to <= 4096 elements it is finished by a parallel
    // scan (one addition per thread and step).  Small windows (<= 4096 buckets) scan their buckets directly; larger ones run
    // ONE chunked level that leaves at most 4096 chunks.
    p.reduce_scan = opt_reduce_scan != 0;
    const uint32_t scan_log = opt_reduce_scan_log ? (uint32_t)opt_reduce_scan_log : 12;
    p.scan_direct = p.reduce_scan && p.half <= (1u << scan_log);
    if (p.reduce_scan && !p.scan_direct) {
      const uint32_t need = ilog2_floor(p.half) - scan_log;   // first-level chunk size that leaves 2^scan_log chunks
      // the chunked level leaves 4096 chunks per bucket window: with the few bucket sets of a table plan (1..6) that is a few thousand
      // lanes walking 512 buckets each on a chip of 65536 -- c = 22 with tables: 9.3 ms of bucket reduction against 3.6 at c = 23,
      // which already took the recursive scheme (profiles/r04_table_levels_sweep.txt)
      if (need > 9 || (need > 8 && p.bucket_windows < 8))
        p.reduce_scan = false;   // windows beyond 2^21 buckets (precomputed tables): keep the recursive scheme and ITS chunk sizes
      else if (p.logL0 < need)
        p.logL0 = need;
    }
    p.L0 = 1u << p.logL0;
    p.T0 = ceil_div(p.half, p.L0);
    p.scan_nb = p.T0;
    if (p.reduce_scan && !p.scan_direct && !opt_reduce_log_chunk && !opt_reduce_log_chunk0) {
      // The first level is one wave per SIMD (an accumulator chain per lane, 53 K lanes): 13 windows x 4096 chunks are 832 waves on
      // the 1024 SIMDs of an MI355X -- a fifth of the chip idles while every lane walks 128 buckets.  Cut the window into as many
      // chunks as fill the SIMDs once (4994 chunks of 105 buckets = 1015 waves) and let the scan run on the next power of two.
      // `reduce_fill` (option, default 1) asks for that many waves per SIMD instead: two waves issue a VALU instruction per ~4.1 cycles
      // where a lone wave issues one per ~5.3, at the price of a scan over twice the elements.
      const uint64_t kSimds = 1024;   // 256 CUs x 4 (the plan is also computed without a device: mi355_msm_plan)
      const uint64_t fill = opt_reduce_fill > 0 ? (uint64_t)opt_reduce_fill : 1;
      const uint64_t waves = ((uint64_t)p.bucket_windows * p.T0 + 63) / 64;
      if (waves < fill * kSimds && p.T0 >= 1024) {
        const uint32_t t_fit = (uint32_t)(fill * kSimds * 64 / p.bucket_windows);
        if (t_fit > p.T0 && t_fit < 2 * fill * p.T0) {
          const uint32_t L = ceil_div(p.half, t_fit);
          if (L >= 8 && L < p.L0) {
            p.L0 = L;
            p.T0 = ceil_div(p.half, L);
            while (p.scan_nb < p.T0) p.scan_nb *= 2;   // the scan runs on the next power of two, the tail of a row stays empty
          }
        }
      }
    }
    return p;
  }
};

namespace {

void ensure_device(mi355_msm_ctx* ctx) { HIP_OK(hipSetDevice(ctx->device)); }

// Every compute entry point starts here: no device, no service (this library has no CPU fallback).  Returns the device count.
int require_device() {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    (void)hipGetLastError();
    throw HipFailure((int)(e != hipSuccess ? e : hipErrorNoDevice), "mi355_msm: no HIP device visible (this library has no CPU fallback)");
  }
  return count;
}

// A stream for host -> device copies that must make progress WHILE the compute stream is full of long kernels.  HIP multiplexes
// streams onto a few hardware queues (GPU_MAX_HW_QUEUES = 4) in creation order; a copy stream that lands on the compute
// stream's queue has its event and barrier packets stuck behind 16-ms kernels -- measured in bench.py, whose stream creation
// order produced exactly that: the stateless pipeline ran at 26 GB/s (346 ms) instead of 54 GB/s (196 ms), and with
// GPU_MAX_HW_QUEUES = 2 or 8 at full speed again.  Queues are pooled per priority, so a high-priority stream never shares
// one with the (normal-priority) compute streams.
hipStream_t create_copy_stream() {
  hipStream_t s = nullptr;
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least &&
      hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest) == hipSuccess)
    return s;
  (void)hipGetLastError();
  HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  return s;
}

// Device bytes of the per-run work buffers of one chunk (keys/vals x2, buckets, slots x2, reduce x4); `el` = 2 for Fq2 points.
uint64_t work_bytes(const Plan& p, uint64_t el, bool carry = false) {
  const PartPlan pp = part_plan((uint32_t)(p.entries / p.windows), p.c, p.windows, p.levels, 0, 0);
  const PartScratchSizes ps = part_scratch_sizes(pp);
  return p.entries * 16 + ps.matrix + ps.partial + ps.segs_a + ps.segs_b + ps.subjob_first + ps.counts + ps.totals +
         (uint64_t)p.bucket_windows * p.half * 224 * el * (carry ? 2 : 1) + 2 * (2ull * p.nlanes) * (224 * el + 4) + (p.scan_direct ? (uint64_t)p.bucket_windows * p.half : 4ull * p.bucket_windows * p.scan_nb) * 224 * el;
}

DevBuf* const* work_buffers(mi355_msm_ctx* ctx, size_t& count) {
  static thread_local DevBuf* bufs[24];
  DevBuf* list[] = {&ctx->entries[0], &ctx->entries[1], &ctx->part_matrix, &ctx->part_partial, &ctx->part_segs[0], &ctx->part_segs[1],
                    &ctx->part_subjobs, &ctx->part_counts, &ctx->part_totals, &ctx->buckets, &ctx->slots[0], &ctx->slots[1],
                    &ctx->slot_keys[0], &ctx->slot_keys[1], &ctx->red_a[0], &ctx->red_a[1], &ctx->red_x[0], &ctx->red_x[1], &ctx->carry_buckets};
  count = sizeof list / sizeof list[0];
  for (size_t i = 0; i < count; i++) bufs[i] = list[i];
  return bufs;
}

// Bytes of each per-chunk work buffer, in the order of work_buffers(): what run_chunk reserves for a chunk of n pairs under
// plan p (`xyzz` = sizeof(XyzzDev) of the curve).  Separate from the reservation so that a caller that knows all its chunk
// sizes in advance (the stateless pipeline) can take the element-wise maximum and allocate ONCE.
struct WorkBytes {
  size_t b[19] = {};
  void max_with(const WorkBytes& o) {
    for (int i = 0; i < 19; i++) b[i] = std::max(b[i], o.b[i]);
  }
};

WorkBytes chunk_work_bytes(const Plan& p, size_t n, bool use_tables, size_t xyzz, bool carry = false) {
  WorkBytes w;
  const PartPlan gp = part_plan((uint32_t)n, p.c, p.windows, use_tables ? p.levels : 1, 0, 0);
  const PartScratchSizes gs = part_scratch_sizes(gp);
  const size_t nbuckets = (size_t)p.bucket_windows * p.half, nslots0 = 2 * (size_t)p.nlanes;
  const size_t red0 = p.scan_direct ? nbuckets : (size_t)p.bucket_windows * p.scan_nb;   // direct scan: a second bucket-sized array to ping-pong with
  w.b[0] = w.b[1] = p.entries * 8 + 64;
  w.b[2] = gs.matrix;
  w.b[3] = gs.partial;
  w.b[4] = gs.segs_a;
  w.b[5] = gs.segs_b;
  w.b[6] = gs.subjob_first;
  w.b[7] = gs.counts;
  w.b[8] = gs.totals;
  w.b[9] = nbuckets * xyzz;
  w.b[10] = w.b[11] = nslots0 * xyzz;
  w.b[12] = w.b[13] = nslots0 * 4;
  w.b[14] = red0 * xyzz;                          // red_a[0]
  w.b[15] = p.scan_direct ? 0 : red0 * xyzz;      // red_a[1]
  w.b[16] = w.b[17] = p.scan_direct ? 0 : red0 * xyzz;   // red_x[0], red_x[1]
  w.b[18] = carry ? nbuckets * xyzz : 0;          // carry_buckets: the same size as `buckets` (the two change places after the first chunk)
  return w;
}

void reserve_work(mi355_msm_ctx* ctx, const WorkBytes& w) {
  size_t nb = 0;
  DevBuf* const* wb = work_buffers(ctx, nb);
  for (size_t i = 0; i < nb; i++)
    if (w.b[i]) wb[i]->reserve(w.b[i]);
}

// The largest chunk (<= want) whose work buffers fit the device memory that is free now or already held by this context
// for the purpose -- the reference plans its allocations before it runs, too (ML msm.cu:453-466).  Halves until it fits.
size_t fit_chunk(mi355_msm_ctx* ctx, size_t want, bool use_tables, uint32_t force_c = 0) {
  size_t free_b = 0, total_b = 0;
  HIP_OK(hipMemGetInfo(&free_b, &total_b));
  size_t held = 0, nb = 0;
  DevBuf* const* wb = work_buffers(ctx, nb);
  for (size_t i = 0; i < nb; i++) held += wb[i]->bytes;
  uint64_t avail = (uint64_t)free_b + held;
  if (ctx->opt_mem_limit > 0 && (uint64_t)ctx->opt_mem_limit < avail) avail = (uint64_t)ctx->opt_mem_limit;
  const uint64_t el = is_g2(ctx->curve) ? 2 : 1;
  size_t cn = want;
  while (cn > 1024) {
    const Plan p = ctx->plan(cn, use_tables, force_c);
    // 3 % head-room for the sort's temporary storage and allocator granularity
    if (p.entries < (1ull << 32) && work_bytes(p, el, force_c != 0) + (work_bytes(p, el, force_c != 0) >> 5) <= avail) break;
    cn = (cn + 1) / 2;
  }
  return cn;
}

// `keep_carry`: a carried batch is under way -- its bucket totals survive (an allocation failure then costs a smaller chunk, not the batch)
void release_work_buffers(mi355_msm_ctx* ctx, bool keep_carry = false) {
  ctx->fitted_chunk = 0;
  size_t nb = 0;
  DevBuf* const* wb = work_buffers(ctx, nb);
  for (size_t i = 0; i < nb; i++)
    if (!(keep_carry && (wb[i] == &ctx->carry_buckets || wb[i] == &ctx->buckets))) wb[i]->release();   // (XYZZ batches total in carry_buckets, Edwards batches in buckets)
}

// Everything the context holds in device memory (the stateless pool's size bound, msm_stateless.hpp).
size_t ctx_device_bytes(mi355_msm_ctx* ctx) {
  size_t nb = 0, total = 0;
  DevBuf* const* wb = work_buffers(ctx, nb);
  for (size_t i = 0; i < nb; i++) total += wb[i]->bytes;
  total += ctx->bases.bytes + ctx->inf.bytes + ctx->scalars.bytes;
  for (const DevBuf& r : ctx->stateless_raw) total += r.bytes;
  return total;
}

// Run `fn.template operator()<Curve>()` for the curve id.
template <class Fn>
void with_curve(int curve, Fn&& fn) {
  switch (curve) {
    case MI355_BLS12_377_G1: fn.template operator()<Bls12_377_G1>(); break;
    case MI355_BLS12_381_G1: fn.template operator()<Bls12_381_G1>(); break;
    case MI355_BLS12_377_G2: fn.template operator()<Bls12_377_G2>(); break;
    case MI355_BLS12_381_G2: fn.template operator()<Bls12_381_G2>(); break;
    default: bad_arg("unknown curve id %d", curve);
  }
}
bool known_curve(int c) { return c >= MI355_BLS12_377_G1 && c <= MI355_BLS12_381_G2; }
size_t coord_bytes(int curve) { return is_g2(curve) ? 96 : 48; }

template <class C>
void convert_bases(mi355_msm_ctx* ctx, const uint8_t* d_raw, size_t n, size_t stride, hipStream_t st) {
  using E = typename C::E;
  using AD = AffineDevT<typename E::T>;
  ctx->bases.reserve(n * sizeof(AD));
  ctx->inf.reserve(n);
  HIP_OK(Launch<E>::convert_bases(d_raw, stride, (uint32_t)n, ctx->bases_serialized, ctx->bases.as<AD>(), ctx->inf.as<uint8_t>(), st));
}

// Build the table levels behind the converted bases (level 0): level j holds 2^(c G j) * P_i, j = 1 .. k-1, where G = ceil(windows / k)
// is the number of bucket sets (k = windows, G = 1: a level per window, the round-1..3 form; CMB PrecomputePoints.cu:10-39 builds
// k = 6 levels 2^(46 j) P for its 23-bit windows, i.e. G = 2).
template <class C>
void build_tables(mi355_msm_ctx* ctx, const uint8_t* d_raw, size_t n, size_t stride, int want_levels, hipStream_t st) {
  using E = typename C::E;
  using El = typename E::T;
  using AD = AffineDevT<El>;
  using XD = XyzzDevT<El>;
  if (ctx->opt_precompute == 2 && ctx->inject_alloc_failures > 0) {   // (test hook: the automatic choice must survive a failed build)
    ctx->inject_alloc_failures--;
    throw HipFailure((int)hipErrorOutOfMemory, "table allocation failed: out of memory (injected by the inject_alloc_failures test hook)");
  }
  const TableShape ts = table_shape(n, C::SCALAR_BITS, ctx->opt_window_bits, want_levels);
  const uint32_t c = ts.c, windows = ts.levels /* = table levels from here on */, bsets = ts.bsets;
  const uint32_t level_shift = c * bsets;   // doublings between two levels
  if ((uint64_t)windows * n >= (1ull << 31)) bad_arg("precompute: %u tables of %zu points exceed the 2^31 index range", windows, n);
  const size_t table_bytes = (size_t)windows * n * sizeof(AD);
  size_t free_b = 0, total_b = 0;
  HIP_OK(hipMemGetInfo(&free_b, &total_b));
  if (table_bytes + n * (sizeof(XD) + sizeof(El)) > free_b + ctx->bases.bytes)
    bad_arg("precompute: %u tables of %zu points need %zu MiB, only %zu MiB free", windows, n, table_bytes >> 20, free_b >> 20);
  ctx->bases.reserve(table_bytes);
  ctx->inf.reserve((size_t)windows * n);
  HIP_OK(Launch<E>::convert_bases(d_raw, stride, (uint32_t)n, ctx->bases_serialized, ctx->bases.as<AD>(), ctx->inf.as<uint8_t>(), st));
  DevBuf xyzz, prefix;
  try {
    xyzz.reserve(n * sizeof(XD));
    prefix.reserve(n * sizeof(El));
    const uint32_t J = 64;
    for (uint32_t w = 1; w < windows; w++) {
      AD* prev = ctx->bases.as<AD>() + (size_t)(w - 1) * n;
      AD* next = ctx->bases.as<AD>() + (size_t)w * n;
      uint8_t* inf_prev = ctx->inf.as<uint8_t>() + (size_t)(w - 1) * n;
      uint8_t* inf_next = ctx->inf.as<uint8_t>() + (size_t)w * n;
      HIP_OK(Launch<E>::pre_double(prev, inf_prev, (uint32_t)n, level_shift, xyzz.as<XD>(), st));
      HIP_OK(Launch<E>::pre_normalize(xyzz.as<XD>(), (uint32_t)n, J, prefix.as<El>(), next, inf_next, st));
    }
    HIP_OK(hipStreamSynchronize(st));
  } catch (...) {
    xyzz.release();
    prefix.release();
    throw;
  }
  xyzz.release();
  prefix.release();
  ctx->pre_c = c;
  ctx->pre_windows = windows;
}

// BLS12-377 G1: twisted-Edwards records for every table level (te.hpp).  The base set takes the fast path only if every
// (non-infinite) entry has an image; otherwise, or when the records do not fit, the context stays on XYZZ.
void build_te(mi355_msm_ctx* ctx, size_t n, hipStream_t st) {
  const size_t levels = ctx->pre_c ? ctx->pre_windows : 1;
  const size_t total = levels * n;
  size_t free_b = 0, total_b = 0;
  HIP_OK(hipMemGetInfo(&free_b, &total_b));
  const size_t need = total * sizeof(TeAffineDev) + n * sizeof(Fe) + ((size_t)1 << 30);
  if (need > free_b + ctx->te_bases.bytes) return;   // not an error: the XYZZ path needs nothing more
  ctx->te_bases.reserve(total * sizeof(TeAffineDev));
  ctx->flags.reserve(2 * sizeof(uint32_t));
  if (!ctx->h_flags) HIP_OK(hipHostMalloc((void**)&ctx->h_flags, 2 * sizeof(uint32_t), hipHostMallocDefault));
  HIP_OK(hipMemsetAsync(ctx->flags.p, 0, 2 * sizeof(uint32_t), st));
  DevBuf prefix;
  try {
    prefix.reserve(n * sizeof(Fe));
    for (size_t w = 0; w < levels; w++)
      HIP_OK(LaunchTe::convert(ctx->bases.as<AffineDev>() + w * n, ctx->inf.as<uint8_t>() + w * n, (uint32_t)n, 64, prefix.as<Fe>(),
                               ctx->te_bases.as<TeAffineDev>() + w * n, ctx->flags.as<uint32_t>(), st));
    HIP_OK(hipMemcpyAsync(ctx->h_flags, ctx->flags.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
  } catch (...) {
    prefix.release();
    throw;
  }
  prefix.release();
  if (ctx->h_flags[0] != 0) {   // a 2-torsion point or one of the two u = -1 points among the bases
    ctx->te_bases.release();
    return;
  }
  ctx->te_active = true;
  if (levels > 1) {
    // keep level 0 of the short-Weierstrass tables only: it serves the (rare) XYZZ fallback, without tables
    DevBuf level0;
    level0.reserve(n * sizeof(AffineDev));
    HIP_OK(hipMemcpyAsync(level0.p, ctx->bases.p, n * sizeof(AffineDev), hipMemcpyDeviceToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
    ctx->bases.release();
    ctx->bases = level0;
    ctx->sw_level0_only = true;
  }
}

// "precompute" = 2: the table levels this base set gets, from the device memory that is free NOW (what the context already holds for
// bases counts as free: it is replaced).  Candidates are the shapes profiles/r04_table_levels_sweep.txt shows as wins at 2^26 pairs -- a
// level per window (-6 %, 151 GB with the Edwards records), 6 levels (-3 %, 86 GB), 4 (-1.4 %, 60 GB), 3 (-1 %, 47 GB); two levels lose --
// tried largest first; each must fit with its build temporaries AND leave the work buffers of a full chunk plus a tenth of the device to
// the caller.  -1 = no tables (small inputs, where tables were never measured to pay, or not enough memory: none under ~64 GB free at 2^26).
int precompute_auto_levels(mi355_msm_ctx* ctx, size_t n) {
  // Below 2^24 pairs (round 6, profiles/r06_size_sweep_tables.txt): where the bucket reduction and the fragment merge are a quarter of
  // an MSM, SIX levels -- a window size of 16..17 bits with 3 bucket sets instead of 17..24 -- pay at least 5 %:
  //   BLS12-377 G1  2^18 -7.5 %, 2^19 -13.7 %, 2^20 -6.1 % (2^21 -3.3 %: left alone, 2^22..2^23: tables lose)
  //   BLS12-381 G1  2^18 -23 %, 2^19 -20 %   (from 2^20 on tables lose)
  //   G2            2^18 -18 %, 2^19 -22 %, 2^20 -15.5 %, 2^21 -7.8 %   (2^22: +-0)
  // for 0.2 .. 3.2 GB of tables and 20 .. 230 ms of (untimed) init.  Nothing was measured below 2^18: no tables there.
  const bool mid = n < ((size_t)1 << 24);
  if (mid) {
    const size_t upper = is_g2(ctx->curve) ? ((size_t)3 << 20) : (is_381(ctx->curve) ? ((size_t)3 << 18) : ((size_t)3 << 19));
    if (n < ((size_t)1 << 18) || n >= upper) return -1;
  }
  size_t free_b = 0, total_b = 0;
  HIP_OK(hipMemGetInfo(&free_b, &total_b));
  size_t held = ctx->bases.bytes + ctx->te_bases.bytes + ctx->inf.bytes, nb = 0;
  DevBuf* const* wb = work_buffers(ctx, nb);
  for (size_t i = 0; i < nb; i++) held += wb[i]->bytes;
  uint64_t avail = (uint64_t)free_b + held;
  const uint64_t reserve = total_b / 10;
  avail = avail > reserve ? avail - reserve : 0;
  if (ctx->opt_mem_limit > 0 && (uint64_t)ctx->opt_mem_limit < avail) avail = (uint64_t)ctx->opt_mem_limit;   // (test hook)
  const uint64_t el = is_g2(ctx->curve) ? 2 : 1;
  const bool te = ctx->curve == MI355_BLS12_377_G1 && ctx->opt_twisted_edwards;
  // candidates, largest first: G1 {all, 6, 4, 3} (each a measured win at 2^26: profiles/r04_table_levels_sweep.txt); G2 {all, 6} only --
  // 3 levels LOSE there (115.5 ms against 111.4 without tables, profiles/r05_g2_tables_and_window.txt)
  static const int kG1[] = {0, 6, 4, 3}, kG2[] = {0, 6}, kMid[] = {6};
  const int* cand = mid ? kMid : (el == 2 ? kG2 : kG1);
  const int ncand = mid ? 1 : (el == 2 ? 2 : 4);
  for (int ci = 0; ci < ncand; ci++) {
    const int want = cand[ci];
    const TableShape ts = table_shape(n, ctx->scalar_bits(), ctx->opt_window_bits, want);
    if (want > 0 && ts.levels > (uint32_t)want) continue;
    if ((uint64_t)ts.levels * n >= (1ull << 31) || ts.levels < 2) continue;
    const uint64_t sw = (uint64_t)ts.levels * n * 128 * el, inf = (uint64_t)ts.levels * n;
    const uint64_t te_b = te ? (uint64_t)ts.levels * n * sizeof(TeAffineDev) : 0;
    // peak of the build: the streamed Edwards build holds three short-Weierstrass levels, every Edwards level and one XYZZ + prefix array
    // (build_tables_te_streamed); the plain build every level and the same temporaries
    const uint64_t build = te ? 3 * n * 128 + te_b + n * (224 + 56) : sw + n * (224 + 56) * el;
    mi355_msm_ctx tmp;   // (planning arithmetic only; the options that shape a plan are the context's)
    tmp.curve = ctx->curve;
    tmp.pre_c = ts.c;
    tmp.pre_windows = ts.levels;
    tmp.opt_window_bits = ctx->opt_window_bits;
    tmp.opt_assume_subgroup = ctx->opt_assume_subgroup;
    tmp.opt_anchor = ctx->opt_anchor;
    tmp.opt_carry = ctx->opt_carry;
    tmp.opt_lane_entries = ctx->opt_lane_entries;
    tmp.opt_seg_entries = ctx->opt_seg_entries;
    tmp.opt_reduce_scan = ctx->opt_reduce_scan;
    tmp.opt_reduce_fill = ctx->opt_reduce_fill;
    const Plan p = tmp.plan(std::min(n, (size_t)1 << 26));
    const uint64_t steady = (te ? n * 128 * el + te_b : sw) + inf + work_bytes(p, el) + n * 32;
    const uint64_t need = std::max(build + inf, steady);
    if (need + (need >> 4) <= avail) return want;
  }
  return -1;
}

// Tables AND their twisted-Edwards records in one sweep (BLS12-377 G1 with the Edwards path on -- the default): level w of the
// short-Weierstrass table exists only to produce level w + 1 and its own Edwards record, so two rotating level buffers (+ level 0, which
// the rare XYZZ fallback keeps) replace the whole table.  build_tables + build_te hold every level of BOTH forms at once -- 95 + 142 GB
// for a level per window at 2^26 -- which is why "precompute" = auto could not afford the best shape on an idle 288-GB device; this
// way the peak is 3 x 8.6 + 142 + 19 GB.  Returns false -- nothing kept -- when a point of some level has no Edwards image (a base set
// off the odd-order subgroup): the caller then builds the plain tables.
bool build_tables_te_streamed(mi355_msm_ctx* ctx, const uint8_t* d_raw, size_t n, size_t stride, int want_levels, hipStream_t st) {
  using C = Bls12_377_G1;
  using E = C::E;
  if (ctx->opt_precompute == 2 && ctx->inject_alloc_failures > 0) {   // (test hook: the automatic choice must survive a failed build)
    ctx->inject_alloc_failures--;
    throw HipFailure((int)hipErrorOutOfMemory, "table allocation failed: out of memory (injected by the inject_alloc_failures test hook)");
  }
  const TableShape ts = table_shape(n, C::SCALAR_BITS, ctx->opt_window_bits, want_levels);
  const uint32_t levels = ts.levels, level_shift = ts.c * ts.bsets;
  if ((uint64_t)levels * n >= (1ull << 31)) bad_arg("precompute: %u tables of %zu points exceed the 2^31 index range", levels, n);
  if (levels < 2) return false;
  const size_t need = (size_t)levels * n * sizeof(TeAffineDev) + 3 * n * sizeof(AffineDev) + n * (sizeof(XyzzDev) + sizeof(Fe)) + (size_t)levels * n;
  size_t free_b = 0, total_b = 0;
  HIP_OK(hipMemGetInfo(&free_b, &total_b));
  if (need > free_b + ctx->bases.bytes + ctx->te_bases.bytes + ctx->inf.bytes)
    bad_arg("precompute: %u table levels of %zu points need %zu MiB, only %zu MiB free", levels, n, need >> 20, free_b >> 20);
  DevBuf xyzz, prefix, level0;   // (all three released on every path out: DevBuf has no destructor)
  bool ok = false;
  try {
    ctx->te_bases.reserve((size_t)levels * n * sizeof(TeAffineDev));
    ctx->bases.reserve(3 * n * sizeof(AffineDev));       // slot 0 = level 0 (kept), slots 1 / 2 = the two most recent levels
    ctx->inf.reserve((size_t)levels * n);
    ctx->flags.reserve(2 * sizeof(uint32_t));
    if (!ctx->h_flags) HIP_OK(hipHostMalloc((void**)&ctx->h_flags, 2 * sizeof(uint32_t), hipHostMallocDefault));
    HIP_OK(hipMemsetAsync(ctx->flags.p, 0, 2 * sizeof(uint32_t), st));
    xyzz.reserve(n * sizeof(XyzzDev));
    prefix.reserve(n * sizeof(Fe));
    AffineDev* const slots = ctx->bases.as<AffineDev>();
    uint8_t* const inf = ctx->inf.as<uint8_t>();
    auto slot_of = [&](uint32_t w) { return slots + (size_t)(w == 0 ? 0 : 1 + ((w - 1) & 1)) * n; };
    HIP_OK(Launch<E>::convert_bases(d_raw, stride, (uint32_t)n, ctx->bases_serialized, slot_of(0), inf, st));
    for (uint32_t w = 0; w < levels; w++) {
      if (w > 0) {
        HIP_OK(Launch<E>::pre_double(slot_of(w - 1), inf + (size_t)(w - 1) * n, (uint32_t)n, level_shift, xyzz.as<XyzzDev>(), st));
        HIP_OK(Launch<E>::pre_normalize(xyzz.as<XyzzDev>(), (uint32_t)n, 64, prefix.as<Fe>(), slot_of(w), inf + (size_t)w * n, st));
      }
      HIP_OK(LaunchTe::convert(slot_of(w), inf + (size_t)w * n, (uint32_t)n, 64, prefix.as<Fe>(), ctx->te_bases.as<TeAffineDev>() + (size_t)w * n,
                               ctx->flags.as<uint32_t>(), st));
    }
    HIP_OK(hipMemcpyAsync(ctx->h_flags, ctx->flags.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    ok = ctx->h_flags[0] == 0;
    if (ok) {
      // keep level 0 of the short-Weierstrass form only: it serves the (rare) XYZZ fallback, without tables
      level0.reserve(n * sizeof(AffineDev));
      HIP_OK(hipMemcpyAsync(level0.p, ctx->bases.p, n * sizeof(AffineDev), hipMemcpyDeviceToDevice, st));
      HIP_OK(hipStreamSynchronize(st));   // (a device-to-device hipMemcpy is not host-synchronous: the source is freed next)
      ctx->bases.release();
      ctx->bases = level0;
      level0 = DevBuf{};   // (ownership moved)
    }
  } catch (...) {
    xyzz.release();
    prefix.release();
    level0.release();
    ctx->te_bases.release();
    throw;
  }
  xyzz.release();
  prefix.release();
  if (!ok) {
    ctx->te_bases.release();
    return false;
  }
  ctx->pre_c = ts.c;
  ctx->pre_windows = levels;
  ctx->te_active = true;
  ctx->sw_level0_only = true;
  return true;
}

void anchor_prepare(mi355_msm_ctx* ctx, size_t n, hipStream_t st);   // (below, with the anchored window's sum of bases)

void set_bases_device(mi355_msm_ctx* ctx, const void* d_affine, size_t n, size_t stride) {
  ensure_device(ctx);
  const size_t min_stride = 2 * coord_bytes(ctx->curve) + (ctx->bases_serialized ? 0 : 1);
  if (stride < min_stride || (stride & 3)) bad_arg("affine stride %zu is not a 4-byte multiple >= %zu", stride, min_stride);
  if (n >= (1ull << 31)) bad_arg("npoints %zu exceeds 2^31-1", n);
  ctx->pre_c = ctx->pre_windows = 0;
  ctx->nbases = 0;
  ctx->anchor_sums.clear();   // sums of the previous bases
  ctx->fitted_chunk = 0;   // the plan (tables, window size) may change with the base set: fit the first chunk again
  ctx->te_active = false;
  ctx->te_fallback_streak = 0;
  ctx->sw_level0_only = false;
  ctx->auto_levels = 0;
  // The buffers of the PREVIOUS base set go back first: DevBuf::reserve only grows, so a second set_bases used to keep the
  // larger of the two allocations (tables of 150 GB behind a table-free context), precompute_auto_levels counted that memory as free
  // although te_bases is reserved while the old bases are still allocated, and "base_bytes" reported the old size.
  ctx->bases.release();
  ctx->te_bases.release();
  ctx->inf.release();
  bool te_done = false;   // the Edwards records were built together with the tables (build_tables_te_streamed)
  if (n) {
    bool tables = ctx->opt_precompute == 1;
    int want_levels = (int)ctx->opt_table_levels;
    if (ctx->opt_precompute == 2) {
      want_levels = precompute_auto_levels(ctx, n);
      tables = want_levels >= 0;
    }
    if (tables) {
      try {
        if (ctx->curve == MI355_BLS12_377_G1 && ctx->opt_twisted_edwards)
          te_done = build_tables_te_streamed(ctx, (const uint8_t*)d_affine, n, stride, want_levels, ctx->own_stream);
        if (!te_done)
          with_curve(ctx->curve, [&]<class C>() { build_tables<C>(ctx, (const uint8_t*)d_affine, n, stride, want_levels, ctx->own_stream); });
      } catch (const HipFailure& e) {
        // auto: tables are an optimisation -- a build that runs out of memory after all (another tenant took it meanwhile) leaves the
        // context on the table-free path instead of failing set_bases
        if (ctx->opt_precompute != 2 || (e.code != (int)hipErrorOutOfMemory && e.code != -1)) throw;
        (void)hipGetLastError();
        ctx->pre_c = ctx->pre_windows = 0;
        ctx->bases.release();
        ctx->inf.release();
        tables = false;
      }
    }
    if (!tables)
      with_curve(ctx->curve, [&]<class C>() { convert_bases<C>(ctx, (const uint8_t*)d_affine, n, stride, ctx->own_stream); });
    if (ctx->opt_precompute == 2 && tables) ctx->auto_levels = ctx->pre_windows;
    HIP_OK(hipStreamSynchronize(ctx->own_stream));
    if (ctx->curve == MI355_BLS12_377_G1 && ctx->opt_twisted_edwards && !te_done) build_te(ctx, n, ctx->own_stream);
  }
  ctx->nbases = n;
  // the sum of all bases, where a run over all of them would use an anchored window: here, in the (untimed) init, not in the first run
  if (n) anchor_prepare(ctx, n, ctx->own_stream);
}

// ---- the host tail (window fold, chunk sums, normalisation) -----------------------------------------------------------
// On 64-bit limbs (host_fold64.hpp, ~4x faster than the device representation run on a CPU core), G1 and G2 alike.
template <class F>
const Fp64& fp64_of() {
  static const Fp64 ctx = [] {
    Fp64 f{};
    f.init<F>();
    if constexpr (std::is_same_v<F, Bls12_377_Fq>) {
      f.const_from_device<F>(f.two_d, Bls12_377_Te::K2D);
      f.const_from_device<F>(f.sqrt3, Bls12_377_Te::SQRT3);
      f.const_from_device<F>(f.fsc_sqrt3, Bls12_377_Te::FSC_SQRT3);
    }
    return f;
  }();
  return ctx;
}

template <class F, int NB>
const Fp2_64<NB>& fp2_64_of() {
  static const Fp2_64<NB> ctx{&fp64_of<F>()};
  return ctx;
}

template <class E>
struct HostTail;
template <class F, int NB>
struct HostTail<Fp2El<F, NB>> {   // G2: the same code over Fp2
  using Pt = XyzzG64<F2_64>;
  static void set_inf(Pt& a) { sw64_set_inf(a); }
  static void add(Pt& a, const Pt& b) { sw64_add(fp2_64_of<F, NB>(), a, b); }
  static void dbl(Pt& a) { sw64_dbl(fp2_64_of<F, NB>(), a); }
  static void fold(Pt& out, const XyzzT<Fe2>* sums, int windows, int c) { fold_windows64<F>(fp2_64_of<F, NB>(), out, sums, windows, c); }
  static void to_abi(uint8_t* out, const Pt& a) { sw64_to_abi(fp2_64_of<F, NB>(), out, a); }
};
template <class F>
struct HostTail<FpEl<F>> {
  using Pt = Xyzz64;
  static void set_inf(Pt& a) { sw64_set_inf(a); }
  static void add(Pt& a, const Pt& b) { sw64_add(fp64_of<F>(), a, b); }
  static void dbl(Pt& a) { sw64_dbl(fp64_of<F>(), a); }
  static void fold(Pt& out, const Xyzz* sums, int windows, int c) { fold_windows64<F>(fp64_of<F>(), out, sums, windows, c); }
  // (the Edwards kernels hold their points in the limb shape of TeFq: te.hpp)
  static bool fold_te(Pt& out, const Xyzz* sums, int windows, int c) {
    if constexpr (std::is_same_v<F, Bls12_377_Fq>)
      return fold_windows_te64<F, TeFq>(fp64_of<F>(), out, sums, windows, c);
    else
      return fold_windows_te64<F>(fp64_of<F>(), out, sums, windows, c);
  }
  static void to_abi(uint8_t* out, const Pt& a) { sw64_to_abi(fp64_of<F>(), out, a); }
};

// The W window sums (the first element of every window's row) straight into pinned host memory, the two flag words with
// them, flag 1 re-armed: one tiny kernel instead of a 2-D copy, a copy and a memset -- the runtime's copy path alone left a
// 22-us hole in the timeline of a small MSM (tools/gap_probe.py).
__global__ void __launch_bounds__(256) k_collect_sums(const uint4* __restrict__ src, uint32_t row_stride_u4, uint32_t row_u4, uint32_t rows,
                                                      uint4* __restrict__ dst_host, uint32_t* __restrict__ flags, uint32_t* __restrict__ flags_host) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < rows * row_u4) {
    const uint32_t r = i / row_u4, c = i - r * row_u4;
    dst_host[i] = src[(size_t)r * row_stride_u4 + c];
  }
  if (flags_host && i == 0) {
    flags_host[0] = flags[0];
    flags_host[1] = flags[1];
    flags[1] = 0;
  }
}

// A batch that runs as several chunks carries ONE bucket array through them (k_bucket_merge): every chunk groups and accumulates
// with the window size of the whole batch (of at most 2^26 pairs of it), chunk 0's buckets become the batch's, the later ones are added to them, and only the
// last chunk reduces, synchronises and folds.  The chunks before it return the identity and leave their kernels in flight.
struct BucketCarry {
  uint32_t c;        // window bits of the batch
  uint32_t index;    // chunk number within the batch (selects its stage events)
  bool first, last;
};

// stage events of chunk `index` of a carried batch (the single-chunk path keeps ctx->ev)
hipEvent_t* carry_events(mi355_msm_ctx* ctx, uint32_t index) {
  while (ctx->carry_ev.size() < 7 * ((size_t)index + 1)) {
    hipEvent_t e = nullptr;
    HIP_OK(hipEventCreate(&e));
    ctx->carry_ev.push_back(e);
  }
  return ctx->carry_ev.data() + 7 * (size_t)index;
}

// One chunk of one batch: device scalars [0, n) against bases [base0, base0 + n).  Leaves the folded chunk sum in `out`.
// TE = true runs the twisted-Edwards kernels (BLS12-377 G1 contexts whose bases all have an image) and returns false when
// an addition reported a vanishing denominator: the caller then repeats the chunk -- or, when `carry` is set, the whole batch -- with TE = false.
template <class C, bool TE>
bool run_chunk_impl(mi355_msm_ctx* ctx, const uint32_t* d_scalars, size_t base0, size_t n, hipStream_t st,
                    typename HostTail<typename C::E>::Pt& out, const std::function<void()>* while_gpu_busy, const BucketCarry* carry = nullptr) {
  using E = typename C::E;
  using El = typename E::T;
  using XyzzDev = XyzzDevT<El>;
  using AffineDev = AffineDevT<El>;
  using SegOut = SegOutT<El>;
  using Xyzz = XyzzT<El>;
  const bool use_tables = ctx->pre_c && (TE || !ctx->sw_level0_only);
  const Plan p = ctx->plan(n, use_tables, carry ? carry->c : 0);
  if (p.entries >= (1ull << 32)) bad_arg("chunk of %zu pairs needs %llu sort entries (>= 2^32)", n, (unsigned long long)p.entries);
  hipEvent_t* const ev = carry ? carry_events(ctx, carry->index) : ctx->ev;
  const size_t NE = p.entries;
  if (ctx->inject_alloc_failures != 0) {
    // N > 0: the next N reservations fail; -K: the K-th reservation from now fails (a chunk in the middle of a carried batch)
    const bool fail = ctx->inject_alloc_failures > 0 ? (ctx->inject_alloc_failures--, true) : (++ctx->inject_alloc_failures == 0);
    if (fail) throw HipFailure((int)hipErrorOutOfMemory, "work-buffer reservation failed: out of memory (injected by the inject_alloc_failures test hook)");
  }
  reserve_work(ctx, chunk_work_bytes(p, n, use_tables, sizeof(XyzzDev), carry != nullptr));
  const uint32_t table_stride = use_tables ? (uint32_t)ctx->nbases : 0u;
  const PartPlan gp = part_plan((uint32_t)n, p.c, p.windows, use_tables ? p.levels : 1, (uint32_t)base0, table_stride, ctx->opt_assume_subgroup != 0, p.anchor);
  ctx->last_anchor = gp.anchor;
  const size_t nbuckets = (size_t)p.bucket_windows * p.half;
  if (ctx->pinned_bytes < p.windows * sizeof(XyzzDev)) {
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned_bytes = 64 * sizeof(XyzzDev) > p.windows * sizeof(XyzzDev) ? 64 * sizeof(XyzzDev) : p.windows * sizeof(XyzzDev);
    HIP_OK(hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocDefault));
  }

  const AffineDev* bases = ctx->bases.as<AffineDev>();
  const uint8_t* inf = ctx->inf.as<uint8_t>();
  uint32_t* flags = ctx->flags.as<uint32_t>();

  // digits + bucket grouping: (value, key) entries sorted by key in entries[sorted]; the count of real entries stays on the device
  HIP_OK(hipEventRecord(ev[0], st));
  PartBuffers gb{};
  gb.entries[0] = ctx->entries[0].as<uint2>();
  gb.entries[1] = ctx->entries[1].as<uint2>();
  gb.matrix = ctx->part_matrix.as<uint32_t>();
  gb.partial = ctx->part_partial.as<uint32_t>();
  gb.segs[0] = ctx->part_segs[0].as<PartSeg>();
  gb.segs[1] = ctx->part_segs[1].as<PartSeg>();
  gb.subjob_first = ctx->part_subjobs.as<uint32_t>();
  gb.counts = ctx->part_counts.as<uint32_t>();
  gb.totals = ctx->part_totals.as<uint32_t>();
  hipError_t gerr = hipSuccess;
  const int sorted = PartLaunch::run((ctx->curve == MI355_BLS12_381_G1 || ctx->curve == MI355_BLS12_381_G2) ? 1 : 0, ctx->opt_scalars_montgomery != 0, d_scalars, inf, gp, gb, st,
                                     ev[1], gerr);
  HIP_OK(gerr);
  const uint2* entries = gb.entries[sorted];
  const uint32_t* n_real = gb.totals;
  // Carried batches: the twisted-Edwards kernels accumulate a later chunk straight onto the buckets the earlier chunks left
  // (SegOutT::carry_in); the XYZZ kernels fill a fresh array that k_bucket_merge adds to the batch's (see carry_begin_run for why).
  const bool carry_in = TE && carry && !carry->first;
  if (!carry_in) HIP_OK(hipMemsetAsync(ctx->buckets.p, 0, nbuckets * sizeof(XyzzDev), st));
  HIP_OK(hipEventRecord(ev[2], st));

  SegOut so{ctx->buckets.as<XyzzDev>(), ctx->slots[0].as<XyzzDev>(), ctx->slot_keys[0].as<uint32_t>()};
  so.carry_in = carry_in ? 1u : 0u;
  if constexpr (TE)
    HIP_OK(LaunchTe::accumulate(entries, n_real, p.K, ctx->te_bases.as<TeAffineDev>(), so, p.nlanes, flags, st));
  else
    HIP_OK(Launch<E>::accumulate(entries, n_real, p.K, bases, so, p.nlanes, st, (ctx->opt_g2_paired & 1) != 0));
  HIP_OK(hipEventRecord(ev[3], st));
#ifdef MSM_DEBUG
  {
    // the debug build stops here after every chunk: the grouping's violation counters and the slot keys the accumulation left
    char what[320];
    HIP_OK(PartLaunch::debug_finish(gp, gb, ctx->slot_keys[0].as<uint32_t>(), 2 * p.nlanes, st, what, sizeof what, &ctx->debug_checks));
    if (what[0]) throw HipFailure(-2, what);
  }
#endif

  // merge the run fragments that crossed lane boundaries
  uint32_t n_in = 2 * p.nlanes;
  int cur = 0;
  if (p.nlanes > 1) {
    for (;;) {
      uint32_t nl = ceil_div(n_in, p.segK);
      if (nl > 1 && 2 * (uint64_t)nl >= n_in) bad_arg("fragment merge would not shrink (%u slots, fan-in %u)", n_in, p.segK);
      SegOut o{ctx->buckets.as<XyzzDev>(), ctx->slots[cur ^ 1].as<XyzzDev>(), ctx->slot_keys[cur ^ 1].as<uint32_t>()};
      if constexpr (TE)
        HIP_OK(LaunchTe::segreduce(ctx->slots[cur].as<XyzzDev>(), ctx->slot_keys[cur].as<uint32_t>(), n_in, p.segK, o, nl, ctx->quad_limit, flags, st));
      else
        HIP_OK(Launch<E>::segreduce(ctx->slots[cur].as<XyzzDev>(), ctx->slot_keys[cur].as<uint32_t>(), n_in, p.segK, o, nl, ctx->quad_limit, st,
                                    (ctx->opt_g2_paired & 4) != 0));
      if (nl == 1) break;
      n_in = 2 * nl;
      cur ^= 1;
    }
  }
  // carried buckets
  XyzzDev* bucket_src = ctx->buckets.as<XyzzDev>();
  if constexpr (!TE) {
    // XYZZ: the first chunk's array becomes the batch's, later chunks are added to it
    if (carry) {
      if (carry->first) {
        std::swap(ctx->buckets, ctx->carry_buckets);
      } else {
        HIP_OK(Launch<E>::bucket_merge(ctx->carry_buckets.as<XyzzDev>(), ctx->buckets.as<XyzzDev>(), (uint32_t)nbuckets, st, (ctx->opt_g2_paired & 16) != 0));
      }
      bucket_src = ctx->carry_buckets.as<XyzzDev>();
    }
  }
  HIP_OK(hipEventRecord(ev[4], st));
  ctx->last_info[0] = p.c;
  ctx->last_info[1] = p.windows;
  ctx->last_info[6] = ctx->pre_c ? 1 : 0;
  ctx->last_info[2] = NE;
  ctx->last_info[3] = p.K;
  ctx->last_info[4] += 1;
  ctx->last_info[5] = p.nlanes;
  ctx->last_info[7] = TE ? 1 : 0;
  ctx->last_bucket_windows = p.bucket_windows;
  ctx->last_l1_bits = gp.hb;
  ctx->last_l1_bins = gp.nbins;
  {
    uint32_t rb_[4];
    ctx->last_passes = (uint32_t)part_pass_bits(gp.lb, rb_);
  }
  if (carry && !carry->last) {
    // nothing to reduce yet, nothing to wait for: the next chunk's kernels queue up behind these
    HIP_OK(hipEventRecord(ev[5], st));
    HIP_OK(hipEventRecord(ev[6], st));
    if (while_gpu_busy) (*while_gpu_busy)();
    HostTail<E>::set_inf(out);
    return true;
  }

  // buckets -> one point per window
  const XyzzDev* sums_src = nullptr;   // where the W window sums end up: element 0 of rows that are sums_stride elements apart
  uint32_t sums_stride = 1;
  if (p.reduce_scan) {
    // parallel scan, one addition per thread and step.  Direct: on the buckets, ping-pong with a second bucket-sized array.
    // Otherwise: one chunked level first (A_t, X_t per chunk), scan on the X_t, join with the A_t, tree.
    uint32_t nb = p.half;
    XyzzDev* bufs[2] = {bucket_src, ctx->red_a[0].as<XyzzDev>()};
    const XyzzDev* a_sums = nullptr;
    if (!p.scan_direct) {
      if (p.scan_nb != p.T0) {
        // rows of scan_nb elements, T0 of them written: the rest must read as empty
        HIP_OK(hipMemsetAsync(ctx->red_a[0].p, 0, (size_t)p.bucket_windows * p.scan_nb * sizeof(XyzzDev), st));
        HIP_OK(hipMemsetAsync(ctx->red_x[0].p, 0, (size_t)p.bucket_windows * p.scan_nb * sizeof(XyzzDev), st));
      }
      if constexpr (TE)
        HIP_OK(LaunchTe::bucket_reduce(true, nullptr, bucket_src, p.half, p.L0, p.T0, p.bucket_windows, p.scan_nb,
                                       ctx->red_a[0].as<XyzzDev>(), ctx->red_x[0].as<XyzzDev>(), flags, st));
      else
        HIP_OK(Launch<E>::bucket_reduce(true, nullptr, bucket_src, p.half, p.L0, p.T0, p.bucket_windows, p.scan_nb,
                                        ctx->red_a[0].as<XyzzDev>(), ctx->red_x[0].as<XyzzDev>(), st, (ctx->opt_g2_paired & 2) != 0));
      nb = p.scan_nb;
      a_sums = ctx->red_a[0].as<XyzzDev>();
      bufs[0] = ctx->red_x[0].as<XyzzDev>();
      bufs[1] = ctx->red_x[1].as<XyzzDev>();
    }
    int cur = 0;
    auto step = [&](uint32_t d, uint32_t mode) {
      if constexpr (TE)
        HIP_OK(LaunchTe::reduce_scan_step(bufs[cur], a_sums, bufs[cur ^ 1], nb, p.bucket_windows, d, mode, ctx->quad_limit, flags, st));
      else
        HIP_OK(Launch<E>::reduce_scan_step(bufs[cur], a_sums, bufs[cur ^ 1], nb, p.bucket_windows, d, mode, ctx->quad_limit, st, (ctx->opt_g2_paired & 8) != 0));
      cur ^= 1;
    };
    for (uint32_t d = 1; d < nb; d <<= 1) step(d, 0);
    if (a_sums) step(0, 2);
    if (nb & (nb - 1)) bad_arg("scan reduction needs a power-of-two element count per window (%u)", nb);
    for (uint32_t h = nb >> 1; h >= 1; h >>= 1) step(h, 1);   // out_j = in_j + in_(j+h) for j < h
    if (nb == 1 && !a_sums) step(1, 0);   // a single bucket per window: one pass that normalises an empty bucket to the identity
    HIP_OK(hipEventRecord(ev[5], st));
    // the window sums sit at the head of each window's row
    sums_src = bufs[cur];
    sums_stride = nb;
  } else {
    uint32_t n_per_win = p.half, logL = p.logL0, chunks = p.T0;
    int rb = 0;
    if constexpr (TE)
      HIP_OK(LaunchTe::bucket_reduce(true, nullptr, bucket_src, n_per_win, 1u << logL, chunks, p.bucket_windows, chunks,
                                     ctx->red_a[0].as<XyzzDev>(), ctx->red_x[0].as<XyzzDev>(), flags, st));
    else
      HIP_OK(Launch<E>::bucket_reduce(true, nullptr, bucket_src, n_per_win, 1u << logL, chunks, p.bucket_windows, chunks,
                                      ctx->red_a[0].as<XyzzDev>(), ctx->red_x[0].as<XyzzDev>(), st, (ctx->opt_g2_paired & 2) != 0));
    while (chunks > 1) {
      n_per_win = chunks;
      logL = p.logL;
      chunks = ceil_div(n_per_win, 1u << logL);
      if constexpr (TE)
        HIP_OK(LaunchTe::bucket_reduce(false, ctx->red_a[rb].as<XyzzDev>(), ctx->red_x[rb].as<XyzzDev>(), n_per_win, 1u << logL, chunks,
                                       p.bucket_windows, chunks, ctx->red_a[rb ^ 1].as<XyzzDev>(), ctx->red_x[rb ^ 1].as<XyzzDev>(), flags, st));
      else
        HIP_OK(Launch<E>::bucket_reduce(false, ctx->red_a[rb].as<XyzzDev>(), ctx->red_x[rb].as<XyzzDev>(), n_per_win, 1u << logL, chunks,
                                        p.bucket_windows, chunks, ctx->red_a[rb ^ 1].as<XyzzDev>(), ctx->red_x[rb ^ 1].as<XyzzDev>(), st, (ctx->opt_g2_paired & 2) != 0));
      rb ^= 1;
    }
    HIP_OK(hipEventRecord(ev[5], st));
    sums_src = ctx->red_a[rb].as<XyzzDev>();
    sums_stride = 1;
  }
  {
    static_assert(sizeof(XyzzDev) % 16 == 0, "window sums are collected in 16-byte pieces");
    const uint32_t row_u4 = sizeof(XyzzDev) / 16, total = p.bucket_windows * row_u4;
    hipLaunchKernelGGL(k_collect_sums, dim3((total + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint4*>(sums_src), sums_stride * row_u4, row_u4,
                       p.bucket_windows, reinterpret_cast<uint4*>(ctx->pinned), TE ? flags : nullptr, TE ? ctx->h_flags : nullptr);
    HIP_OK(hipGetLastError());
  }
  HIP_OK(hipEventRecord(ev[6], st));
  // everything for this chunk is enqueued: host work that should hide behind it (the next batch's H2D copy) goes here
  if (while_gpu_busy) (*while_gpu_busy)();
  // A small MSM is ~0.5 ms of device time and the blocking wait of the runtime wakes up in steps of ~0.15 ms (wall times of
  // 2^10..2^16 pairs clustered at 0.66 / 0.81 / 0.96 / 1.12 ms): poll the chunk's last event for the first few milliseconds,
  // then block as before.
  if (p.entries < (1ull << 26)) {
    const auto t_spin = std::chrono::steady_clock::now();
    while (hipEventQuery(ev[6]) == hipErrorNotReady) {
      if (std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(4)) break;
    }
    (void)hipGetLastError();
  }
  HIP_OK(hipStreamSynchronize(st));

  typename E::Md md;
  std::vector<Xyzz> sums(p.bucket_windows);
  const XyzzDev* hs = reinterpret_cast<const XyzzDev*>(ctx->pinned);
  for (uint32_t w = 0; w < p.bucket_windows; w++) sums[w] = hs[w].p;
  bool ok = true;
  const auto t_fold = std::chrono::steady_clock::now();
  if constexpr (TE)
    ok = ctx->h_flags[1] == 0 && HostTail<E>::fold_te(out, sums.data(), (int)p.bucket_windows, (int)p.c);
  else
    HostTail<E>::fold(out, sums.data(), (int)p.bucket_windows, (int)p.c);
  ctx->last_ms[MI355_T_HOST_FOLD] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_fold).count();

  // stage times: of this chunk, or of every chunk of the carried batch (their events have all completed by now)
  for (uint32_t k = 0, nk = carry ? carry->index + 1 : 1; k < nk; k++) {
    hipEvent_t* const e = carry ? carry_events(ctx, k) : ctx->ev;
    float ms = 0;
    for (int s = 0; s < 5; s++) {
      HIP_OK(hipEventElapsedTime(&ms, e[s], e[s + 1]));
      ctx->last_ms[s] += ms;
    }
    HIP_OK(hipEventElapsedTime(&ms, e[0], e[6]));
    ctx->last_ms[MI355_T_TOTAL] += ms;
  }
  return ok;
}

// A base set that trips the incomplete law twice in a row (points outside the prime-order subgroup) would pay for both paths
// on every call: demote the context to XYZZ for good and give the twisted-Edwards records back.
void te_fell_back(mi355_msm_ctx* ctx) {
  if (++ctx->te_fallback_streak >= 2) {
    ctx->te_active = false;
    ctx->te_demotions++;
    ctx->te_bases.release();
  }
}

// Returns false only for a chunk of a CARRIED batch whose twisted-Edwards run failed: the batch's buckets are Edwards sums then, so
// the caller repeats the whole batch with allow_te = false (a lone chunk is repeated here, on the XYZZ path).
template <class C>
bool run_chunk(mi355_msm_ctx* ctx, const uint32_t* d_scalars, size_t base0, size_t n, hipStream_t st,
               typename HostTail<typename C::E>::Pt& out, const std::function<void()>* while_gpu_busy = nullptr,
               const BucketCarry* carry = nullptr, bool allow_te = true) {
  if constexpr (std::is_same_v<C, Bls12_377_G1>) {
    if (ctx->te_active && allow_te) {
      if (run_chunk_impl<C, true>(ctx, d_scalars, base0, n, st, out, while_gpu_busy, carry)) {
        if (!carry || carry->last) ctx->te_fallback_streak = 0;
        return true;
      }
      ctx->te_fallbacks++;
      if (carry) return false;
      // (the hook -- the next batch's H2D copy -- has run already)
      run_chunk_impl<C, false>(ctx, d_scalars, base0, n, st, out, nullptr);
      te_fell_back(ctx);
      return true;
    }
  }
  run_chunk_impl<C, false>(ctx, d_scalars, base0, n, st, out, while_gpu_busy, carry);
  return true;
}

// ---- anchored window: the sum of the bases of a run, and its multiples (see choose_window_bits) ------------------------------------
// S = the sum of bases [0, n) (the ones not flagged infinite): k_sum_bases leaves one fragment per lane of 32..512 bases, the fragment merge
// of the pipeline adds them up into bucket 0, k_collect_sums brings it (and the Edwards failure flag) to the host, the host tail turns it
// into a point.  ~n mixed additions: 7 ms at 2^26.  TE = true returns false when an addition hit a vanishing denominator (bases outside
// the prime-order subgroup): the caller repeats on the XYZZ records.
template <class C, bool TE>
bool sum_bases_impl(mi355_msm_ctx* ctx, size_t n, hipStream_t st, typename HostTail<typename C::E>::Pt& S) {
  using E = typename C::E;
  using El = typename E::T;
  using XyzzDev = XyzzDevT<El>;
  using SegOut = SegOutT<El>;
  constexpr uint32_t kSegK = 8;
  const uint32_t kPerLane = (uint32_t)std::min<size_t>(512, std::max<size_t>(32, n >> 17));   // ~2^17 lanes: a chip's worth, whatever n
  const uint32_t nlanes = ceil_div(n, kPerLane);
  ctx->buckets.reserve(sizeof(XyzzDev));
  for (int k = 0; k < 2; k++) {
    ctx->slots[k].reserve(2 * (size_t)nlanes * sizeof(XyzzDev));
    ctx->slot_keys[k].reserve(2 * (size_t)nlanes * sizeof(uint32_t));
  }
  if (ctx->pinned_bytes < 64 * sizeof(XyzzDev)) {
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned_bytes = 64 * sizeof(XyzzDev);
    HIP_OK(hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocDefault));
  }
  uint32_t* flags = ctx->flags.as<uint32_t>();
  HIP_OK(hipMemsetAsync(ctx->buckets.p, 0, sizeof(XyzzDev), st));   // the empty-bucket marker: no base at all
  SegOut so{ctx->buckets.as<XyzzDev>(), ctx->slots[0].as<XyzzDev>(), ctx->slot_keys[0].as<uint32_t>()};
  if constexpr (TE)
    HIP_OK(LaunchTe::sum_bases(ctx->te_bases.as<TeAffineDev>(), ctx->inf.as<uint8_t>(), 0, (uint32_t)n, kPerLane, so, nlanes, flags, st));
  else
    HIP_OK(Launch<E>::sum_bases(ctx->bases.as<AffineDevT<El>>(), ctx->inf.as<uint8_t>(), 0, (uint32_t)n, kPerLane, so, nlanes, st));
  uint32_t n_in = 2 * nlanes;
  for (int cur = 0;; cur ^= 1) {
    const uint32_t nl = ceil_div(n_in, kSegK);
    SegOut o{ctx->buckets.as<XyzzDev>(), ctx->slots[cur ^ 1].as<XyzzDev>(), ctx->slot_keys[cur ^ 1].as<uint32_t>()};
    if constexpr (TE)
      HIP_OK(LaunchTe::segreduce(ctx->slots[cur].as<XyzzDev>(), ctx->slot_keys[cur].as<uint32_t>(), n_in, kSegK, o, nl, ctx->quad_limit, flags, st));
    else
      HIP_OK(Launch<E>::segreduce(ctx->slots[cur].as<XyzzDev>(), ctx->slot_keys[cur].as<uint32_t>(), n_in, kSegK, o, nl, ctx->quad_limit, st, false));
    if (nl == 1) break;
    n_in = 2 * nl;
  }
  const uint32_t row_u4 = sizeof(XyzzDev) / 16;
  hipLaunchKernelGGL(k_collect_sums, dim3((row_u4 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint4*>(ctx->buckets.p), row_u4, row_u4, 1u,
                     reinterpret_cast<uint4*>(ctx->pinned), TE ? flags : nullptr, TE ? ctx->h_flags : nullptr);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(st));
  const XyzzDev* hs = reinterpret_cast<const XyzzDev*>(ctx->pinned);
  bool empty = true;
  for (size_t k = 0; k < sizeof(XyzzDev); k++) empty = empty && reinterpret_cast<const uint8_t*>(hs)[k] == 0;
  if (empty) {
    HostTail<E>::set_inf(S);
    return !TE || ctx->h_flags[1] == 0;
  }
  const XyzzT<El> sum = hs[0].p;
  if constexpr (TE)
    return ctx->h_flags[1] == 0 && HostTail<E>::fold_te(S, &sum, 1, 1);
  else
    HostTail<E>::fold(S, &sum, 1, 1);
  return true;
}

// Kept per context and n until the bases change.
template <class C>
void anchor_sum_of_bases(mi355_msm_ctx* ctx, size_t n, hipStream_t st, typename HostTail<typename C::E>::Pt& S) {
  using E = typename C::E;
  using Pt = typename HostTail<E>::Pt;
  static_assert(std::is_trivially_copyable_v<Pt>, "cached as bytes");
  for (const auto& a : ctx->anchor_sums)
    if (a.n == n && a.shift < 0) {
      memcpy(&S, a.pt.data(), sizeof S);
      return;
    }
  const auto t0 = std::chrono::steady_clock::now();
  bool done = false;
  if constexpr (std::is_same_v<C, Bls12_377_G1>) {
    if (ctx->te_active) done = sum_bases_impl<C, true>(ctx, n, st, S);   // (a failure here is not counted as a fallback of a run)
  }
  if (!done) sum_bases_impl<C, false>(ctx, n, st, S);
  if (ctx->anchor_sums.size() >= 12) ctx->anchor_sums.erase(ctx->anchor_sums.begin(), ctx->anchor_sums.begin() + 4);
  mi355_msm_ctx::AnchorSum a{n, -1, std::vector<uint8_t>(sizeof S)};
  memcpy(a.pt.data(), &S, sizeof S);
  ctx->anchor_sums.push_back(std::move(a));
  ctx->anchor_sums_computed++;
  ctx->anchor_sum_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// Whether a run over bases [0, n) would use an anchored window, and if so its sum of bases (from the cache, or computed now).  Short of
// memory for the handful of slots that takes: false, and the run uses plain digits.
template <class C>
bool anchor_ready(mi355_msm_ctx* ctx, size_t n, hipStream_t st, typename HostTail<typename C::E>::Pt& S) {
  if (!ctx->anchor_wanted(n)) return false;
  const bool tables0 = ctx->pre_c && (ctx->te_active || !ctx->sw_level0_only);
  if (anchor_window((int)ctx->plan(std::min(n, (size_t)1 << 26), tables0).c, ctx->scalar_bits(), ctx->opt_anchor == 2) == kNoAnchor) return false;
  try {
    anchor_sum_of_bases<C>(ctx, n, st, S);
    return true;
  } catch (const HipFailure& e) {
    if (e.code != (int)hipErrorOutOfMemory) throw;
    (void)hipStreamSynchronize(st);
    release_work_buffers(ctx);
    return false;
  }
}
void anchor_prepare(mi355_msm_ctx* ctx, size_t n, hipStream_t st) {
  with_curve(ctx->curve, [&]<class C>() {
    typename HostTail<typename C::E>::Pt S;
    (void)anchor_ready<C>(ctx, n, st, S);
  });
  ctx->anchor_sum_ms = 0;
}

// 2^shift x S (shift ~ 250 host doublings, ~0.15 ms: kept beside S)
template <class C>
void anchor_term(mi355_msm_ctx* ctx, size_t n, int shift, const typename HostTail<typename C::E>::Pt& S, typename HostTail<typename C::E>::Pt& out) {
  using E = typename C::E;
  for (const auto& a : ctx->anchor_sums)
    if (a.n == n && a.shift == shift) {
      memcpy(&out, a.pt.data(), sizeof out);
      return;
    }
  out = S;
  for (int i = 0; i < shift; i++) HostTail<E>::dbl(out);
  mi355_msm_ctx::AnchorSum a{n, shift, std::vector<uint8_t>(sizeof out)};
  memcpy(a.pt.data(), &out, sizeof out);
  ctx->anchor_sums.push_back(std::move(a));
}

// Streams the scalar batches of a host-pointer run: batch b+1 is copied while batch b computes
// (the reference's double-buffered batches, P1A 6block/cuda/pippenger_inf.cu:110-160; CMB MSM.cu:419-505), and the FIRST
// batch -- whose copy nothing can hide -- is handed over in two pieces, 1/4 then 3/4, so that the first quarter is already
// accumulating while the rest crosses PCIe (the reference's quarter-split first copy, CMB MSM.cu:419-434).
// `host_batch_bytes` is the distance between two batches in the CALLER's buffer: a shard of a sharded context reads its
// slice out of batches that are npoints_total scalars apart.
struct HostBatches {
  const uint8_t* host;
  uint8_t* dev;
  size_t batch_bytes;        // bytes of one batch on the device (n * 32)
  size_t host_batch_bytes;   // bytes between batches in the host buffer
  hipStream_t copy_stream;
  hipEvent_t ready[2];   // whole batch b resident: ready[b & 1]
  hipEvent_t piece[7];   // batch 0 arrives in up to eight pieces: piece i resident (the last piece signals ready[0])
  void copy_pairs(size_t b, size_t first, size_t count, hipEvent_t ev) const {
    HIP_OK(hipMemcpyAsync(dev + b * batch_bytes + first * 32, host + b * host_batch_bytes + first * 32, count * 32, hipMemcpyHostToDevice,
                          copy_stream));
    HIP_OK(hipEventRecord(ev, copy_stream));
  }
  void copy(size_t b) const { copy_pairs(b, 0, batch_bytes / 32, ready[b & 1]); }
};

// `dev_batch_pairs`: distance (in scalars) between two batches in the device buffer (n, or the total of a sharded run when a
// shard reads its slice in place).
template <class C>
void run_device_t(mi355_msm_ctx* ctx, uint8_t* out, const uint32_t* d_scalars, size_t n, size_t batches, size_t dev_batch_pairs,
                  hipStream_t st, const HostBatches* hb) {
  using E = typename C::E;
  // An allocation failure caps the chunks of THIS run only (one transient event -- fragmentation, another tenant's spike --
  // must not cost every later MSM an extra bucket reduction): the next run plans against the memory that is free then.
  size_t max_chunk = ctx->opt_max_chunk ? (size_t)ctx->opt_max_chunk : ((size_t)1 << 26);
  ctx->chunk_cap = 0;
  const size_t out_bytes = 3 * 4 * E::WORDS;
  // Anchored window: where the batch's window size has one (see choose_window_bits), the sum of bases [0, n) must be at hand before
  // anything of this run is in flight -- computed now if this context has not run n pairs before.  Short of memory for that: plain digits.
  typename HostTail<E>::Pt anchor_S;
  struct AnchorScope {
    mi355_msm_ctx* c;
    ~AnchorScope() { c->anchor_armed = false; }
  } anchor_scope{ctx};
  ctx->anchor_armed = false;
  ctx->anchor_sum_ms = 0;
  if (batches) ctx->anchor_armed = anchor_ready<C>(ctx, n, st, anchor_S);
  // first piece of a host-scalar batch = 1/div of it.  Carried, with a merge pass per piece (XYZZ): 13 (1/13 + 3/13 + 9/13); carried onto
  // the stored buckets (twisted Edwards, round 6: a piece costs no merge): 26, a fourth piece and half the PCIe wait before the first
  // kernel (2^26: 112.1 -> 110.2 ms same-box, profiles/r06_ab_carry_in.txt); not carried: 4
  const size_t div = ctx->opt_first_piece_div ? (size_t)ctx->opt_first_piece_div : (ctx->opt_carry ? (ctx->te_active ? 26 : 13) : 4);
  const std::vector<size_t> pb = (hb && batches) ? msm_host::first_batch_pieces(n, max_chunk, div) : std::vector<size_t>{0, n};
  const size_t P = pb.size() - 1;   // pieces of batch 0
  size_t issued = 0;                // ... whose copy has been issued
  std::vector<char> awaited(P, 0);
  auto piece_event = [&](size_t i) { return i + 1 == P ? hb->ready[0] : hb->piece[i]; };
  const std::function<void()> issue_next_piece = [&] {
    hb->copy_pairs(0, pb[issued], pb[issued + 1] - pb[issued], piece_event(issued));
    issued++;
  };
  if (hb && batches && n) issue_next_piece();
  for (size_t b = 0; b < batches; b++) {
    typename HostTail<E>::Pt total;
    HostTail<E>::set_inf(total);
    const bool split = hb && b == 0 && P > 1;
    if (hb && n) {
      HIP_OK(hipStreamWaitEvent(st, b == 0 ? piece_event(0) : hb->ready[b & 1], 0));
      if (b == 0) awaited[0] = 1;
    }
    bool prefetched = false;
    const std::function<void()> prefetch = [&] {
      if (hb && b + 1 < batches) hb->copy(b + 1);
      prefetched = true;
    };
    // A batch that runs as several chunks carries one bucket array through them (BucketCarry): decided at its first chunk
    BucketCarry carry{};
    bool carried = false, allow_te = true;
    uint32_t batch_anchor = kNoAnchor, batch_c = 0;   // the anchored window of the batch's chunks (all of them, or none)
    size_t reclaimed_at = (size_t)-1;   // chunk position at which the idle stateless contexts were last reclaimed
    for (size_t off = 0; off < n;) {
      // plan the chunk against the memory that is there (ML msm.cu:453-466 plans first, too) ...
      const bool tables_now = ctx->pre_c && ((ctx->te_active && allow_te) || !ctx->sw_level0_only);
      size_t cn = std::min(max_chunk, n - off);
      size_t piece = 0;   // the piece of batch 0 this chunk lies in: a chunk never crosses into a piece that may not have arrived
      if (split) {
        while (off >= pb[piece + 1]) piece++;
        cn = std::min(cn, pb[piece + 1] - off);
      }
      auto fit = [&] {
        const uint32_t fc = carried ? carry.c : 0;
        if (cn > ctx->fitted_chunk || tables_now != ctx->fitted_tables || fc != ctx->fitted_c) cn = fit_chunk(ctx, cn, tables_now, fc);
      };
      fit();
      if (off == 0 && !carried && cn < n && ctx->opt_carry) {
        carried = true;
        // the window size of the whole batch -- of at most 2^26 pairs of it: beyond that the model's choice (c = 22) pays a third
        // grouping pass and a 25-M-bucket merge per chunk (2^28 on one context: 447 ms against 439 at c = 20, tools/big_carry_probe.py)
        carry.c = ctx->plan(std::min(n, (size_t)1 << 26), tables_now).c;
        carry.index = 0;
        fit();                                  // ... under which a chunk needs other buffers
        if (split && !ctx->opt_mem_limit) {
          // The pieces of the first host-scalar batch GROW (1/13, 3/13, 9/13): sized chunk by chunk, every larger piece would free
          // and re-allocate the entry / slot / grouping buffers on a context's first run, and hipFree is a device synchronisation
          // while the previous piece's kernels are still queued.  One reservation for the largest need of any piece.
          WorkBytes w;
          bool fits = true;
          for (size_t i = 0; i < P; i++) {
            const size_t cnt = std::min(max_chunk, pb[i + 1] - pb[i]);
            if (!cnt) continue;
            const Plan pl = ctx->plan(cnt, tables_now, carry.c);
            fits = fits && pl.entries < (1ull << 32);
            w.max_with(chunk_work_bytes(pl, cnt, tables_now, sizeof(XyzzDevT<typename E::T>), true));
          }
          if (fits) {
            try {
              reserve_work(ctx, w);
            } catch (const HipFailure& e) {
              if (e.code != (int)hipErrorOutOfMemory) throw;
              release_work_buffers(ctx);   // short of memory: the per-chunk fit / back-off below sizes the chunks
              fit();
            }
          }
        }
      }
      const bool last = off + cn >= n;
      if (split && !awaited[piece]) {
        while (issued <= piece) issue_next_piece();
        HIP_OK(hipStreamWaitEvent(st, piece_event(piece), 0));
        awaited[piece] = 1;
      }
      typename HostTail<E>::Pt part;
      const std::function<void()>* hook = (split && issued < P) ? &issue_next_piece : ((last && !prefetched) ? &prefetch : nullptr);
      carry.first = off == 0;
      carry.last = last;
      bool ok = true;
      try {
        ok = run_chunk<C>(ctx, d_scalars + (b * dev_batch_pairs + off) * 8, off, cn, st, part, hook, carried ? &carry : nullptr, allow_te);
      } catch (const HipFailure& e) {
        // ... and if an allocation fails all the same (fragmentation, another tenant), give the work buffers back and go on
        // with half the chunk; results do not depend on the chunking (the totals of a carried batch are kept)
        if (e.code != (int)hipErrorOutOfMemory || cn <= 1024) throw;
        (void)hipStreamSynchronize(st);
        release_work_buffers(ctx, carried && off > 0);
        // memory this library still holds behind the caller's back (parked stateless contexts) goes first: same chunk again
        // (once per chunk position: concurrent callers that keep parking contexts must not keep this loop spinning -- ADVICE r4)
        if (reclaimed_at != off && reclaim_idle_device_memory()) {
          reclaimed_at = off;
          continue;
        }
        max_chunk = ctx->chunk_cap = (cn + 1) / 2;   // for the rest of this run
        ctx->oom_backoffs++;
        continue;
      }
      if (!ok) {
        // the twisted-Edwards run of a carried batch failed (flagged at its last chunk): the whole batch again, on the XYZZ path
        allow_te = false;
        off = 0;
        carry.index = 0;
        ctx->fitted_chunk = 0;
        continue;
      }
      const uint32_t fc = carried ? carry.c : 0;
      if (cn > ctx->fitted_chunk || tables_now != ctx->fitted_tables || fc != ctx->fitted_c) {
        ctx->fitted_chunk = cn;
        ctx->fitted_tables = tables_now;
        ctx->fitted_c = fc;
      }
      if (off == 0) {
        batch_anchor = ctx->last_anchor;
        batch_c = (uint32_t)ctx->last_info[0];
      } else if (ctx->last_anchor != batch_anchor || (batch_anchor != kNoAnchor && (uint32_t)ctx->last_info[0] != batch_c)) {
        throw HipFailure(-2, "mi355_msm: the chunks of a batch disagree about its anchored window (internal error)");
      }
      HostTail<E>::add(total, part);
      off += cn;
      carry.index++;
    }
    if (batch_anchor != kNoAnchor) {
      // every scalar's digit in the anchored window was taken relative to 2^(c-1): add 2^(c a + c - 1) x (sum of the bases)
      typename HostTail<E>::Pt term;
      anchor_term<C>(ctx, n, (int)(batch_c * batch_anchor + batch_c - 1), anchor_S, term);
      HostTail<E>::add(total, term);
    }
    if (!allow_te) te_fell_back(ctx);
    if (!prefetched) prefetch();   // n == 0
    const auto t0 = std::chrono::steady_clock::now();
    HostTail<E>::to_abi(out + b * out_bytes, total);
    ctx->last_ms[MI355_T_HOST_FOLD] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
}

void run_device(mi355_msm_ctx* ctx, void* out, const void* d_scalars, size_t n, size_t batches, size_t dev_batch_pairs, hipStream_t st,
                const HostBatches* hb = nullptr) {
  ensure_device(ctx);
  if (n > ctx->nbases) bad_arg("npoints %zu exceeds the %zu uploaded bases", n, ctx->nbases);
  if (!out) bad_arg("null output pointer");
  memset(ctx->last_ms, 0, sizeof ctx->last_ms);
  memset(ctx->last_info, 0, sizeof ctx->last_info);
  with_curve(ctx->curve, [&]<class C>() { run_device_t<C>(ctx, (uint8_t*)out, (const uint32_t*)d_scalars, n, batches, dev_batch_pairs, st, hb); });
}

template <class C>
void fold_t(uint8_t* out, const uint8_t* in, size_t count) {
  using E = typename C::E;
  typename E::Md md;
  XyzzT<typename E::T> total;
  xyzz_set_inf<E>(total);
  for (size_t i = 0; i < count; i++) {
    XyzzT<typename E::T> p;
    xyzz_from_projective_abi<E>(p, in + (size_t)3 * 4 * E::WORDS * i, md);
    xyzz_add<E>(total, p, md);
  }
  xyzz_to_projective_abi<E>(out, total, md);
}

// ---- single-device entry bodies (shared by the C ABI below and by the shards of a sharded context) -------------------

void set_bases_host(mi355_msm_ctx* ctx, const void* affine, size_t npoints, size_t stride, bool serialized) {
  ensure_device(ctx);
  DevBuf raw;
  ctx->bases_serialized = serialized;
  try {
    if (npoints) {
      raw.reserve(npoints * stride);
      // (on the context's stream, like everything that follows: nothing here relies on what a null-stream copy does or does not order
      //  against a non-blocking stream -- the lesson of the staging race in msm_sharded.hpp)
      HIP_OK(hipMemcpyAsync(raw.p, affine, npoints * stride, hipMemcpyHostToDevice, ctx->own_stream));
    }
    set_bases_device(ctx, raw.p, npoints, stride);
  } catch (...) {
    ctx->bases_serialized = false;
    raw.release();
    throw;
  }
  ctx->bases_serialized = false;
  raw.release();
}

// Scalars in host memory: `batches` vectors of n scalars, `host_batch_pairs` scalars apart in the caller's buffer.
void run_host(mi355_msm_ctx* ctx, void* out, const void* scalars, size_t n, size_t batches, size_t host_batch_pairs) {
  ensure_device(ctx);
  const size_t bytes = n * batches * 32;
  ctx->scalars.reserve(bytes ? bytes : 32);
  if (!ctx->copy_stream) {
    ctx->copy_stream = create_copy_stream();
    for (auto& ev : ctx->copy_ev) HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  HostBatches hb{(const uint8_t*)scalars, ctx->scalars.as<uint8_t>(), n * 32, host_batch_pairs * 32, ctx->copy_stream,
                 {ctx->copy_ev[0], ctx->copy_ev[1]}, {}};
  for (int i = 0; i < 7; i++) hb.piece[i] = ctx->copy_ev[2 + i];
  try {
    run_device(ctx, out, ctx->scalars.p, n, batches, n, ctx->own_stream, &hb);
  } catch (...) {
    // copies out of the caller's buffer may still be in flight: the caller is free to release it once we return
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(ctx->own_stream);
    throw;
  }
}

}  // namespace

#include "msm_sharded.hpp"
#include "msm_stateless.hpp"

// ---- stream-ordered runs (mi355_msm_run_async) ---------------------------------------------------------------------------------
// What the second-place entry's API offers a prover that overlaps MSMs with its other kernels (ML bellman-cuda.h:48-75: a stream,
// events, host callbacks; used at P1A matter-labs/src/lib.rs:150-190): the call returns at once, the MSM is ordered after the work
// already in the caller's stream, and completion is a callback.  An MSM ends in host arithmetic (the window fold) and has host-side
// decisions on the way (out-of-memory back-off, the XYZZ repeat of an Edwards run that reported a vanishing denominator), none of which
// may run inside a HIP host callback; so each context owns ONE worker thread that executes its jobs in submission order through the very
// code path of mi355_msm_run_device, on the context's own stream, which first waits for an event recorded in the caller's stream.
struct mi355_msm_job {
  mi355_msm_ctx* ctx = nullptr;
  void* out = nullptr;
  const void* d_scalars = nullptr;
  size_t npoints = 0, batches = 0;
  hipEvent_t ready = nullptr;      // recorded in the caller's stream at submission: the scalars are valid from here on
  mi355_msm_done_fn done = nullptr;
  void* user = nullptr;
  std::mutex mu;
  std::condition_variable cv;
  bool finished = false;
  int code = 0;
  std::string message;
};

struct AsyncWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<mi355_msm_job*> queue;
  bool stop = false;
  uint64_t submitted = 0, completed = 0;
};

namespace {

void async_worker_main(mi355_msm_ctx* ctx) {
  AsyncWorker* w = ctx->async;
  for (;;) {
    mi355_msm_job* job = nullptr;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return w->stop || !w->queue.empty(); });
      if (w->queue.empty()) return;   // stop, and nothing left to run
      job = w->queue.front();
      w->queue.pop_front();
    }
    RustError e = guarded_dev([&] {
      (void)require_device();
      if (ctx->shards.empty()) {
        ensure_device(ctx);
        // ordered after the caller's stream WITHOUT blocking a host thread on it: the context's stream waits for the event
        HIP_OK(hipStreamWaitEvent(ctx->own_stream, job->ready, 0));
        run_device(ctx, job->out, job->d_scalars, job->npoints, job->batches, job->npoints, ctx->own_stream);
      } else {
        HIP_OK(hipEventSynchronize(job->ready));   // the shards run on their own devices and streams
        sharded_run(ctx, job->out, job->d_scalars, job->npoints, job->batches, true, nullptr);
      }
    });
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->completed++;
    }
    // the callback sees the finished job state; its RustError is the callback's to free (rusterror.h convention)
    {
      std::lock_guard<std::mutex> lk(job->mu);
      job->code = e.code;
      job->message = e.message ? e.message : "";
    }
    if (job->done) {
      RustError cb{e.code, e.message ? strdup(e.message) : nullptr};
      job->done(job->user, cb);
    }
    if (e.message) free(e.message);
    {
      std::lock_guard<std::mutex> lk(job->mu);
      job->finished = true;
    }
    job->cv.notify_all();
  }
}

void async_shutdown(mi355_msm_ctx* ctx) {
  AsyncWorker* w = ctx->async;
  if (!w) return;
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->stop = true;
  }
  w->cv.notify_all();
  if (w->th.joinable()) w->th.join();   // (pending jobs are run to completion first: their callers hold pointers to them)
  delete w;
  ctx->async = nullptr;
}

}  // namespace

extern "C" {

RustError mi355_msm_run_async(mi355_msm_ctx* ctx, void* out, const void* d_scalars, size_t npoints, size_t batches, void* stream,
                              mi355_msm_done_fn done, void* user, mi355_msm_job** job_out) {
  return guarded_dev([&] {
    if (!ctx) bad_arg("null context");
    if (!out) bad_arg("null output pointer");
    if (!job_out && !done) bad_arg("neither a job handle nor a completion callback was asked for: the result could never be awaited");
    if (npoints * batches && !d_scalars) bad_arg("null scalars pointer");
    if (job_out) *job_out = nullptr;
    (void)require_device();
    if (ctx->shards.empty()) {
      ensure_device(ctx);
      if (npoints > ctx->nbases) bad_arg("npoints %zu exceeds the %zu uploaded bases", npoints, ctx->nbases);
    }
    std::unique_ptr<mi355_msm_job> job(new mi355_msm_job());
    job->ctx = ctx;
    job->out = out;
    job->d_scalars = d_scalars;
    job->npoints = npoints;
    job->batches = batches;
    job->done = done;
    job->user = user;
    HIP_OK(hipEventCreateWithFlags(&job->ready, hipEventDisableTiming));
    hipError_t er = hipEventRecord(job->ready, (hipStream_t)stream);
    if (er != hipSuccess) {
      (void)hipEventDestroy(job->ready);
      HIP_OK(er);
    }
    if (!ctx->async) {
      ctx->async = new AsyncWorker();
      ctx->async->th = std::thread(async_worker_main, ctx);
    }
    mi355_msm_job* raw = job.release();
    {
      std::lock_guard<std::mutex> lk(ctx->async->mu);
      ctx->async->queue.push_back(raw);
      ctx->async->submitted++;
    }
    ctx->async->cv.notify_one();
    if (job_out) {
      *job_out = raw;
    } else {
      // fire-and-forget (callback only): the job object is released by a watcher once it has finished
      std::thread([raw] {
        {
          std::unique_lock<std::mutex> lk(raw->mu);
          raw->cv.wait(lk, [&] { return raw->finished; });
        }
        (void)hipEventDestroy(raw->ready);
        delete raw;
      }).detach();
    }
  });
}

int mi355_msm_job_done(mi355_msm_job* job) {
  if (!job) return 1;
  std::lock_guard<std::mutex> lk(job->mu);
  return job->finished ? 1 : 0;
}

RustError mi355_msm_job_wait(mi355_msm_job* job) {
  if (!job) return fail(-1, "null job");
  {
    std::unique_lock<std::mutex> lk(job->mu);
    job->cv.wait(lk, [&] { return job->finished; });
  }
  RustError e = job->code ? fail(job->code, job->message.c_str()) : ok();
  (void)hipEventDestroy(job->ready);
  delete job;
  return e;
}

}  // extern "C"

extern "C" {

RustError mi355_msm_create(mi355_msm_ctx** out, int curve, int device) {
  return guarded_dev([&] {
    if (!out) bad_arg("null context out-pointer");
    *out = nullptr;
    if (!known_curve(curve)) bad_arg("unknown curve id %d", curve);
    const int count = require_device();
    if (device < 0) HIP_OK(hipGetDevice(&device));
    if (device >= count) bad_arg("device %d out of range (%d visible)", device, count);
    HIP_OK(hipSetDevice(device));
    mi355_msm_ctx* ctx = new mi355_msm_ctx();
    ctx->curve = curve;
    ctx->device = device;
    try {
      HIP_OK(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
      for (auto& ev : ctx->ev) HIP_OK(hipEventCreate(&ev));
    } catch (...) {
      RustError d = mi355_msm_destroy(ctx);
      if (d.message) free(d.message);
      throw;
    }
    *out = ctx;
  });
}

RustError mi355_msm_destroy(mi355_msm_ctx* ctx) {
  if (ctx) async_shutdown(ctx);   // (outside the guard: joins the worker, which runs pending jobs to completion)
  return guarded_dev([&] {
    if (!ctx) return;
    if (!ctx->shards.empty()) {
      sharded_destroy(ctx);
      delete ctx;
      return;
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->own_stream);
    DevBuf* bufs[] = {&ctx->bases, &ctx->inf, &ctx->scalars, &ctx->te_bases, &ctx->flags, &ctx->stateless_raw[0], &ctx->stateless_raw[1], &ctx->stateless_raw[2]};
    for (DevBuf* b : bufs) b->release();
    release_work_buffers(ctx);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
    for (auto& ev : ctx->ev)
      if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : ctx->carry_ev)
      if (ev) (void)hipEventDestroy(ev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    for (auto& ev : ctx->copy_ev)
      if (ev) (void)hipEventDestroy(ev);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    delete ctx;
  });
}

RustError mi355_msm_create_sharded(mi355_msm_ctx** out, int curve, const int* devices, int ndevices) {
  return guarded_dev([&] {
    if (!out) bad_arg("null context out-pointer");
    *out = nullptr;
    if (!known_curve(curve)) bad_arg("unknown curve id %d", curve);
    if (!devices || ndevices < 1 || ndevices > 64) bad_arg("sharded context needs 1..64 devices");
    *out = sharded_create(curve, devices, ndevices);
  });
}

static RustError create_env_devices(mi355_msm_ctx** out, int curve) {
  // MI355_MSM_DEVICES = "0,1,2,3" | "all" | unset: a harness that never heard of more than one GPU picks them up here
  const char* env = getenv("MI355_MSM_DEVICES");
  if (!env || !*env) return mi355_msm_create(out, curve, -1);
  std::vector<int> devs;
  try {
    devs = parse_device_list(env);
  } catch (const std::exception& e) {
    if (out) *out = nullptr;
    return fail(-1, e.what());
  }
  if (devs.size() == 1) return mi355_msm_create(out, curve, devs[0]);
  return mi355_msm_create_sharded(out, curve, devs.data(), (int)devs.size());
}

RustError mi355_msm_create_env(mi355_msm_ctx** out, int curve) {
  RustError e = create_env_devices(out, curve);
  if (e.code || !out || !*out) return e;
  // MI355_MSM_ASSUME_SUBGROUP = 0 | 1: the harness's bases are (are not) all in the order-r subgroup -- option "assume_subgroup"
  const char* sub = getenv("MI355_MSM_ASSUME_SUBGROUP");
  if (sub && *sub) e = mi355_msm_set_option(*out, "assume_subgroup", atol(sub) != 0);
  // MI355_MSM_PRECOMPUTE = auto | 0 | 1 (+ MI355_MSM_TABLE_LEVELS = k): the harness's init is untimed (CMB MSM.cu:380-383 builds its
  // tables there), so a harness that owns the GPU may let the context spend free HBM on tables -- options "precompute" / "table_levels"
  const char* pre = getenv("MI355_MSM_PRECOMPUTE");
  if (!e.code && pre && *pre) {
    // auto | 0 | 1, case-insensitive; anything else is refused rather than guessed ("AUTO" used to mean 0 and "2" to mean 1)
    long v = -1;
    if (strcasecmp(pre, "auto") == 0) v = 2;
    else if (strcmp(pre, "0") == 0 || strcasecmp(pre, "off") == 0) v = 0;
    else if (strcmp(pre, "1") == 0 || strcasecmp(pre, "on") == 0) v = 1;
    e = v >= 0 ? mi355_msm_set_option(*out, "precompute", v) : fail(-1, "MI355_MSM_PRECOMPUTE must be auto, 0 or 1");
  }
  const char* lv = getenv("MI355_MSM_TABLE_LEVELS");
  if (!e.code && lv && *lv) e = mi355_msm_set_option(*out, "table_levels", atol(lv));
  if (e.code) {
    RustError d = mi355_msm_destroy(*out);
    if (d.message) free(d.message);
    *out = nullptr;
  }
  return e;
}

RustError mi355_msm_set_bases(mi355_msm_ctx* ctx, const void* affine, size_t npoints, size_t stride) {
  return guarded_dev([&] {
    if (!ctx) bad_arg("null context");
    if (npoints && !affine) bad_arg("null bases pointer");
    if (!ctx->shards.empty()) return sharded_set_bases(ctx, affine, npoints, stride, false, false);
    set_bases_host(ctx, affine, npoints, stride, false);
  });
}

RustError mi355_msm_set_bases_serialized(mi355_msm_ctx* ctx, const void* records, size_t npoints) {
  return guarded_dev([&] {
    if (!ctx) bad_arg("null context");
    if (npoints && !records) bad_arg("null records pointer");
    const size_t stride = 2 * coord_bytes(ctx->curve);
    if (!ctx->shards.empty()) return sharded_set_bases(ctx, records, npoints, stride, true, false);
    set_bases_host(ctx, records, npoints, stride, true);
  });
}

RustError mi355_msm_point_to_serialized(int curve, const void* projective, void* out_record) {
  return guarded([&] {
    if (!projective || !out_record) bad_arg("null pointer");
    with_curve(curve, [&]<class C>() {
      using E = typename C::E;
      typename E::Md md;
      constexpr int CB = 4 * E::WORDS;
      XyzzT<typename E::T> p;
      xyzz_from_projective_abi<E>(p, (const uint8_t*)projective, md);
      uint8_t* out = (uint8_t*)out_record;
      memset(out, 0, 2 * CB);
      if (xyzz_is_inf<E>(p)) {
        // arkworks writes GroupAffine::zero() = (0, 1, infinity) -- x = 0, y = 1 in normal form (for G2: y.c0 = 1) -- and ORs
        // SWFlags::Infinity into the last byte (P1B nickray driver/algebra/ec/src/models/short_weierstrass_jacobian.rs:154-156,
        // 827-835)
        out[CB] = 1;
        out[2 * CB - 1] |= 0x40;
        return;
      }
      typename E::T t, ti, zzi, zzzi, x, y;
      E::mul(t, p.zz, p.zzz, md);
      el_inv(ti, t, md, (E*)nullptr);
      E::mul(zzi, ti, p.zzz, md);
      E::mul(zzzi, ti, p.zz, md);
      E::mul(x, p.x, zzi, md);
      E::mul(y, p.y, zzzi, md);
      uint32_t w[2 * E::WORDS];
      E::to_plain(w, x, md);
      E::to_plain(w + E::WORDS, y, md);
      memcpy(out, w, 2 * CB);
    });
  });
}

RustError mi355_msm_set_bases_device(mi355_msm_ctx* ctx, const void* d_affine, size_t npoints, size_t stride) {
  return guarded_dev([&] {
    if (!ctx) bad_arg("null context");
    if (npoints && !d_affine) bad_arg("null bases pointer");
    if (!ctx->shards.empty()) return sharded_set_bases(ctx, d_affine, npoints, stride, false, true);
    // the producer (e.g. torch) may have written the buffer on another stream: make it visible first
    ensure_device(ctx);
    HIP_OK(hipDeviceSynchronize());
    set_bases_device(ctx, d_affine, npoints, stride);
  });
}

RustError mi355_msm_run(mi355_msm_ctx* ctx, void* out, const void* scalars, size_t npoints, size_t batches) {
  return guarded_dev([&] {
    if (!ctx) bad_arg("null context");
    if (npoints * batches && !scalars) bad_arg("null scalars pointer");
    if (!ctx->shards.empty()) return sharded_run(ctx, out, scalars, npoints, batches, false, nullptr);
    run_host(ctx, out, scalars, npoints, batches, npoints);
  });
}

RustError mi355_msm_run_device(mi355_msm_ctx* ctx, void* out, const void* d_scalars, size_t npoints, size_t batches,
                               void* stream) {
  return guarded_dev([&] {
    if (!ctx) bad_arg("null context");
    if (npoints * batches && !d_scalars) bad_arg("null scalars pointer");
    if (!ctx->shards.empty()) return sharded_run(ctx, out, d_scalars, npoints, batches, true, (hipStream_t)stream);
    run_device(ctx, out, d_scalars, npoints, batches, npoints, (hipStream_t)stream);
  });
}

RustError mi355_msm_set_option(mi355_msm_ctx* ctx, const char* key, long value) {
  return guarded([&] {
    if (!ctx || !key) bad_arg("null argument");
    std::string k(key);
    if (k == "force_peer_staging") {   // test hook of a sharded context (msm_sharded.hpp); lives in the parent
      ctx->opt_force_peer_staging = value != 0;
      return;
    }
    if (!ctx->shards.empty() && k != "combine") {
      for (mi355_msm_ctx* sh : ctx->shards) {
        RustError e = mi355_msm_set_option(sh, key, value);
        if (e.code) {
          std::string m = e.message ? e.message : "";
          free(e.message);
          throw HipFailure(e.code, m);
        }
      }
      return;
    }
    ctx->fitted_chunk = 0;   // options change the plan, hence the buffer sizes
    if (k == "window_bits") {
      if (value != 0 && (value < 2 || value > 24)) bad_arg("window_bits %ld out of range [2, 24]", value);
      if (ctx->pre_c && value != 0 && (uint32_t)value != ctx->pre_c)
        bad_arg("window_bits is fixed at %u by the precomputed tables of this context", ctx->pre_c);
      ctx->opt_window_bits = value;
    } else if (k == "lane_entries") {
      if (value < 0 || value > (1 << 20)) bad_arg("lane_entries %ld out of range", value);
      ctx->opt_lane_entries = value;
    } else if (k == "max_chunk") {
      if (value < 0 || value > (1L << 27)) bad_arg("max_chunk %ld out of range [1, 2^27]", value);
      ctx->opt_max_chunk = value;
    } else if (k == "seg_entries") {
      // fan-in K maps n slots to 2*ceil(n/K); that only shrinks for K >= 4
      if (value != 0 && (value < 4 || value > 4096)) bad_arg("seg_entries %ld out of range [4, 4096]", value);
      ctx->opt_seg_entries = value;
    } else if (k == "reduce_log_chunk") {
      if (value < 0 || value > 7) bad_arg("reduce_log_chunk %ld out of range [1, 7]", value);
      ctx->opt_reduce_log_chunk = value;
    } else if (k == "reduce_scan_log") {
      if (value != 0 && (value < 6 || value > 18)) bad_arg("reduce_scan_log %ld out of range [6, 18]", value);
      ctx->opt_reduce_scan_log = value;
    } else if (k == "quad_limit") {
      // launches of at most this many additions run four lanes per addition (merge / scan steps); a field of THIS context
      if (value < 0 || value > (1L << 24)) bad_arg("quad_limit %ld out of range [0, 2^24]", value);
      ctx->quad_limit = (uint32_t)value;
    } else if (k == "g2_paired") {
      // G2 contexts: bit mask of the throughput kernels that run two lanes per point (ignored on G1)
      if (value < 0 || value > 31) bad_arg("g2_paired %ld out of range [0, 31]", value);
      ctx->opt_g2_paired = value;
    } else if (k == "reduce_scan") {
      if (value < -1 || value > 1) bad_arg("reduce_scan %ld out of range [-1, 1]", value);
      ctx->opt_reduce_scan = value;
    } else if (k == "reduce_fill") {
      if (value < 0 || value > 4) bad_arg("reduce_fill %ld out of range [0, 4]", value);
      ctx->opt_reduce_fill = value;
    } else if (k == "reduce_log_chunk0") {
      if (value < 0 || value > 7) bad_arg("reduce_log_chunk0 %ld out of range [1, 7]", value);
      ctx->opt_reduce_log_chunk0 = value;
    } else if (k == "twisted_edwards") {
      ctx->opt_twisted_edwards = value != 0;   // takes effect at the next set_bases
    } else if (k == "precompute") {
      // 0 none; 1 tables ("table_levels" of them); 2 = auto: as many levels as the HBM that is free at set_bases pays for, none when
      // that is short (precompute_auto_levels).  Takes effect at the next set_bases.
      if (value < 0 || value > 2) bad_arg("precompute %ld out of range [0, 2]", value);
      ctx->opt_precompute = value;
    } else if (k == "table_levels") {
      // with "precompute": how many table levels to build (0 = one per window, every window then shares ONE bucket set).  With k levels
      // the windows g, g + G, g + 2G, ... share bucket set g (G = ceil(windows / k)): k times the base memory instead of `windows` times
      // -- yrrid's shape is k = 6, two bucket sets (CMB PrecomputePoints.cu:10-39, MSM.cu:380-383).  Takes effect at the next set_bases.
      if (value < 0 || value > 64) bad_arg("table_levels %ld out of range [0, 64]", value);
      ctx->opt_table_levels = value;
    } else if (k == "scalars_montgomery") {
      ctx->opt_scalars_montgomery = value != 0;
    } else if (k == "assume_subgroup") {
      ctx->opt_assume_subgroup = value != 0;
    } else if (k == "carry") {
      ctx->opt_carry = value != 0;
    } else if (k == "anchor") {
      // 1 (default): batches of 2^20 pairs and more end the signed-digit carry chain at the last full window where the window size
      // leaves (almost) no bits above it, and the constant part rides on the sum of the bases (choose_window_bits); 0: plain digits
      // (2: a test setting -- any batch size, any window size, gain or not)
      if (value < 0 || value > 2) bad_arg("anchor %ld out of range [0, 2]", value);
      ctx->opt_anchor = value;
    } else if (k == "first_piece_div") {
      if (value != 0 && (value < 2 || value > 64)) bad_arg("first_piece_div %ld out of range [2, 64]", value);
      ctx->opt_first_piece_div = value;
    } else if (k == "combine") {
      if (value < 0 || value > 2) bad_arg("combine %ld out of range [0, 2]", value);
      ctx->opt_combine = value;
    } else if (k == "mem_limit") {
      // test hook: size chunks as if at most `value` bytes of device memory were available for the work buffers
      if (value < 0) bad_arg("mem_limit must be >= 0");
      ctx->opt_mem_limit = value;
      ctx->chunk_cap = 0;
    } else if (k == "inject_alloc_failures") {
      // test hook: the next `value` work-buffer reservations of this context (every shard of a sharded one) fail with hipErrorOutOfMemory
      ctx->inject_alloc_failures = value;
    } else {
      bad_arg("unknown option '%s'", key);
    }
  });
}

RustError mi355_msm_last_timings(mi355_msm_ctx* ctx, float* ms, uint64_t* info) {
  return guarded([&] {
    if (!ctx) bad_arg("null context");
    if (!ctx->shards.empty()) {
      // shards run concurrently: report the slowest shard per stage, and the plan of shard 0
      memset(ctx->last_ms, 0, sizeof ctx->last_ms);
      for (mi355_msm_ctx* sh : ctx->shards)
        for (int i = 0; i < MI355_T_COUNT; i++) ctx->last_ms[i] = std::max(ctx->last_ms[i], sh->last_ms[i]);
      memcpy(ctx->last_info, ctx->shards[0]->last_info, sizeof ctx->last_info);
    }
    if (ms) memcpy(ms, ctx->last_ms, sizeof ctx->last_ms);
    if (info) memcpy(info, ctx->last_info, sizeof ctx->last_info);
  });
}

RustError mi355_msm_shard_timings(mi355_msm_ctx* ctx, int shard, float* ms, uint64_t* info) {
  return guarded([&] {
    if (!ctx) bad_arg("null context");
    if (ctx->shards.empty()) {
      if (shard != 0) bad_arg("shard %d of an unsharded context", shard);
      if (ms) memcpy(ms, ctx->last_ms, sizeof ctx->last_ms);
      if (info) memcpy(info, ctx->last_info, sizeof ctx->last_info);
      return;
    }
    if (shard < 0 || (size_t)shard >= ctx->shards.size()) bad_arg("shard %d of %zu", shard, ctx->shards.size());
    const mi355_msm_ctx* sh = ctx->shards[(size_t)shard];
    if (ms) memcpy(ms, sh->last_ms, sizeof sh->last_ms);
    if (info) memcpy(info, sh->last_info, sizeof sh->last_info);
  });
}

RustError mi355_msm_query(mi355_msm_ctx* ctx, const char* key, uint64_t* value) {
  return guarded([&] {
    if (!ctx || !key || !value) bad_arg("null argument");
    std::string k(key);
    if (k == "shards") {
      *value = ctx->shards.size();
      return;
    }
    if (k == "rccl_exchanges") {
      *value = ctx->rccl_exchanges;
      return;
    }
    if (k == "peer_stagings") {
      *value = ctx->peer_stagings;
      return;
    }
    if (k == "async_pending") {   // jobs of mi355_msm_run_async submitted and not yet finished
      uint64_t v = 0;
      if (ctx->async) {
        std::lock_guard<std::mutex> lk(ctx->async->mu);
        v = ctx->async->submitted - ctx->async->completed;
      }
      *value = v;
      return;
    }
    if (!ctx->shards.empty()) {
      // counters add up over the shards; "twisted_edwards" is 1 only if every shard runs on that path
      uint64_t sum = 0, all = 1;
      for (mi355_msm_ctx* sh : ctx->shards) {
        uint64_t v = 0;
        RustError e = mi355_msm_query(sh, key, &v);
        if (e.code) {
          std::string m = e.message ? e.message : "";
          free(e.message);
          throw HipFailure(e.code, m);
        }
        sum += v;
        all &= (v != 0);
      }
      // (option-like and geometry keys are not counters -- every shard holds the same option, and a geometry key reports
      //  the per-shard value (the largest, should the slices differ), not the sum over the shards)
      static const char* const kAll[] = {"twisted_edwards", "assume_subgroup", "carry"};
      static const char* const kMean[] = {"table_levels", "table_window_bits"};
      static const char* const kFirst[] = {"precompute", "g2_paired", "guard_tail", "te_limb_bits", "anchor"};
      static const char* const kMax[] = {"bucket_windows", "l1_bits", "l1_bins", "group_passes", "anchored_window", "anchor_sum_us"};
      auto in = [&](const char* const* set, size_t n) {
        for (size_t i = 0; i < n; i++)
          if (k == set[i]) return true;
        return false;
      };
      if (in(kAll, 3)) {
        *value = all;
      } else if (in(kMean, 2)) {
        *value = sum / ctx->shards.size();
      } else if (in(kFirst, 5) || in(kMax, 6)) {
        uint64_t first = 0, mx = 0;
        for (size_t g = 0; g < ctx->shards.size(); g++) {
          uint64_t v = 0;
          RustError e = mi355_msm_query(ctx->shards[g], key, &v);
          if (e.code) {
            std::string m = e.message ? e.message : "";
            free(e.message);
            throw HipFailure(e.code, m);
          }
          if (g == 0) first = v;
          mx = std::max(mx, v);
        }
        *value = in(kFirst, 5) ? first : mx;
      } else {
        *value = sum;
      }
      return;
    }
    if (k == "twisted_edwards")
      *value = ctx->te_active ? 1 : 0;
    else if (k == "twisted_edwards_fallbacks")
      *value = ctx->te_fallbacks;
    else if (k == "twisted_edwards_demotions")
      *value = ctx->te_demotions;
    else if (k == "assume_subgroup")
      *value = ctx->opt_assume_subgroup ? 1 : 0;
    else if (k == "carry")
      *value = ctx->opt_carry ? 1 : 0;
    else if (k == "anchor")
      *value = (uint64_t)ctx->opt_anchor;
    else if (k == "anchored_window")   // of the most recent chunk: 1 + the window whose digits were taken relative to 2^(c-1); 0 = plain digits
      *value = ctx->last_anchor == kNoAnchor ? 0 : (uint64_t)ctx->last_anchor + 1;
    else if (k == "sorted_entries") {  // non-zero digits of the most recent chunk = its mixed additions (read back from the device: synchronises)
      uint32_t v = 0;
      if (ctx->part_totals.p) {
        ensure_device(ctx);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(&v, ctx->part_totals.p, sizeof v, hipMemcpyDeviceToHost));
      }
      *value = v;
    } else if (k == "anchor_sums")       // sums of bases computed for anchored windows since the context was created
      *value = ctx->anchor_sums_computed;
    else if (k == "anchor_sum_us")     // host wall time the most recent run spent on such a sum (0: it was at hand)
      *value = (uint64_t)(ctx->anchor_sum_ms * 1000.0f);
    else if (k == "oom_backoffs")
      *value = ctx->oom_backoffs;
    else if (k == "debug_checks")   // invariant checks a -DMSM_DEBUG build has run on this context (always 0 in the product build)
      *value = ctx->debug_checks;
    else if (k == "chunk_cap")
      *value = ctx->chunk_cap;
    else if (k == "device")
      *value = (uint64_t)ctx->device;
    else if (k == "bases")
      *value = ctx->nbases;
    else if (k == "table_levels")
      *value = ctx->pre_c ? ctx->pre_windows : (ctx->nbases ? 1 : 0);
    else if (k == "table_window_bits")
      *value = ctx->pre_c;
    else if (k == "base_bytes")
      *value = ctx->bases.bytes + ctx->te_bases.bytes + ctx->inf.bytes;
    else if (k == "bucket_windows")   // of the most recent chunk: bucket sets the reduction walks (= windows without tables)
      *value = ctx->last_bucket_windows;
    else if (k == "l1_bits")          // ... bucket bits level 1 of the grouping resolved, its bins, and the generic passes behind it
      *value = ctx->last_l1_bits;
    else if (k == "l1_bins")
      *value = ctx->last_l1_bins;
    else if (k == "group_passes")
      *value = ctx->last_passes;
    else if (k == "precompute")       // 0 none, 1 tables as asked for, 2 = auto (see "table_levels" for what it chose)
      *value = (uint64_t)ctx->opt_precompute;
    else if (k == "g2_paired")
      *value = (uint64_t)ctx->opt_g2_paired;
    else if (k == "guard_tail")       // 1: MI355_MSM_GUARD_TAIL is on -- every device buffer ends at an unmapped page (DevBuf)
      *value = guard_tail_mode() ? 1 : 0;
    else if (k == "te_limb_bits")     // bits per limb of the twisted-Edwards kernels' field (29: 13 x 29 limbs, 28: 14 x 28; csrc/te.hpp TeFq)
      *value = (uint64_t)TeFq::B;
    else
      bad_arg("unknown query '%s'", key);
  });
}

RustError mi355_msm(int curve, void* out, const void* affine, size_t npoints, const void* scalars, size_t ffi_affine_sz) {
  // both operands stream out of the caller's (pageable) memory while earlier slices compute: msm_stateless.hpp
  return guarded_dev([&] {
    if (!known_curve(curve)) bad_arg("unknown curve id %d", curve);
    if (!out) bad_arg("null output pointer");
    (void)require_device();
    // MI355_MSM_DEVICES = "0,1,2,3" | "0-7" | "all" | unset: several GPUs -> one pipeline per shard over its slice of both operands
    std::vector<int> devs{-1};
    const char* env = getenv("MI355_MSM_DEVICES");
    if (env && *env) {
      try {
        devs = parse_device_list(env);
      } catch (const std::exception& e) {
        throw HipFailure(-1, e.what());
      }
    }
    const size_t G = devs.size(), pb = 3 * coord_bytes(curve);
    // MI355_MSM_ASSUME_SUBGROUP applies here as it does to mi355_msm_create_env (a pooled context keeps no option of an earlier call)
    const char* sub_env = getenv("MI355_MSM_ASSUME_SUBGROUP");
    const long assume_sub = (sub_env && *sub_env && atol(sub_env) != 0) ? 1 : 0;
    if (G == 1) {
      StatelessLease ws(curve, devs[0]);
      take(mi355_msm_set_option(ws.ctx, "assume_subgroup", assume_sub));
      stateless_run(ws.ctx, out, affine, npoints, scalars, ffi_affine_sz);
      ws.keep();
      return;
    }
    std::vector<uint8_t> partials(G * pb);
    std::vector<std::string> errors(G);
    std::vector<int> codes(G, 0);
    std::vector<StatelessStats> stats(G);
    std::vector<std::thread> threads;
    for (size_t g = 0; g < G; g++)
      threads.emplace_back([&, g] {
        try {
          size_t lo, hi;
          shard_bounds(npoints, G, g, lo, hi);
          StatelessLease ws(curve, devs[g]);
          take(mi355_msm_set_option(ws.ctx, "assume_subgroup", assume_sub));
          stateless_run(ws.ctx, partials.data() + g * pb, (const uint8_t*)affine + lo * ffi_affine_sz, hi - lo, (const uint8_t*)scalars + lo * 32,
                        ffi_affine_sz);
          ws.keep();
          stats[g] = g_last_stateless;
        } catch (const HipFailure& e) {
          codes[g] = e.code ? e.code : -1;
          errors[g] = e.what();
        } catch (const std::exception& e) {
          codes[g] = -1;
          errors[g] = e.what();
        }
      });
    for (auto& t : threads) t.join();
    for (size_t g = 0; g < G; g++)
      if (codes[g]) throw HipFailure(codes[g], "shard " + std::to_string(g) + " (device " + std::to_string(devs[g]) + "): " + errors[g]);
    take(mi355_msm_fold(curve, out, partials.data(), G));
    g_last_stateless = *std::max_element(stats.begin(), stats.end(), [](const StatelessStats& a, const StatelessStats& b) { return a.total_ms < b.total_ms; });
  });
}

RustError mi355_msm_last_stateless(double* out, size_t count) {
  return guarded([&] {
    if (!out) bad_arg("null output");
    const StatelessStats& s = g_last_stateless;
    const double v[10] = {s.total_ms, s.setup_ms, s.wait_upload_ms, s.compute_ms, s.tail_ms, s.slices, s.threads, s.bytes, s.dma_done_ms, s.first_dma_ms};
    for (size_t i = 0; i < count && i < 10; i++) out[i] = v[i];
  });
}

RustError mi355_msm_trim(void) {
  return guarded_dev([&] {
    std::vector<StageRing*> rings;
    {
      std::lock_guard<std::mutex> lk(g_ring_mu);
      rings.swap(g_rings_idle);
    }
    for (StageRing* r : rings) {
      (void)hipSetDevice(r->device);
      ring_destroy(r);
    }
    (void)reclaim_idle_device_memory();
  });
}

RustError mi355_msm_pool_stats(uint64_t* out, size_t count) {
  return guarded([&] {
    if (!out) bad_arg("null output");
    uint64_t v[4] = {0, 0, 0, 0};
    {
      std::lock_guard<std::mutex> lk(g_ring_mu);
      v[0] = g_stateless_idle.size();
      for (mi355_msm_ctx* c : g_stateless_idle) v[1] += ctx_device_bytes(c);
      v[2] = g_rings_idle.size();
      v[3] = v[2] * StageRing::SLOTS * StageRing::PIECE;
    }
    for (size_t i = 0; i < count && i < 4; i++) out[i] = v[i];
  });
}

RustError mi355_msm_fold(int curve, void* out, const void* projective, size_t count) {
  return guarded([&] {
    if (!out || (count && !projective)) bad_arg("null pointer");
    with_curve(curve, [&]<class C>() { fold_t<C>((uint8_t*)out, (const uint8_t*)projective, count); });
  });
}

RustError mi355_msm_generate_points(int curve, uint64_t seed, size_t distinct, size_t npoints, void* out, size_t stride) {
  return guarded([&] {
    if (npoints && !out) bad_arg("null output pointer");
    if (!known_curve(curve)) bad_arg("unknown curve id %d", curve);
    if (stride < 2 * coord_bytes(curve) + 1) bad_arg("affine stride %zu too small", stride);
    with_curve(curve, [&]<class C>() { generate_points<C>(seed, distinct, npoints, (uint8_t*)out, stride); });
  });
}

RustError mi355_msm_plan(int curve, size_t npoints, int precompute, const long* options, uint64_t* out) {
  return guarded([&] {
    if (!known_curve(curve)) bad_arg("unknown curve id %d", curve);
    if (!out) bad_arg("null output");
    mi355_msm_ctx tmp;   // never touches the device: only the planning arithmetic
    tmp.curve = curve;
    if (options) {
      tmp.opt_window_bits = options[0];
      tmp.opt_lane_entries = options[1];
      tmp.opt_seg_entries = options[2];
    }
    if (tmp.opt_window_bits && (tmp.opt_window_bits < 2 || tmp.opt_window_bits > 24)) bad_arg("window_bits out of range");
    if (tmp.opt_seg_entries && tmp.opt_seg_entries < 4) bad_arg("seg_entries out of range");
    if (precompute < 0) bad_arg("precompute must be >= 0");
    if (precompute) {
      // 1 = a table level per window, k > 1 = the context option "table_levels" = k: the very shape build_tables gives the context
      const TableShape ts = table_shape(npoints, tmp.scalar_bits(), tmp.opt_window_bits, precompute > 1 ? precompute : 0);
      tmp.pre_c = ts.c;
      tmp.pre_windows = ts.levels;
    }
    const Plan p = tmp.plan(npoints);
    // fragment-merge levels: n slots -> 2*ceil(n/segK) until one lane is left
    uint64_t merge_levels = 0;
    if (p.nlanes > 1) {
      uint32_t n_in = 2 * p.nlanes;
      for (;;) {
        uint32_t nl = ceil_div(n_in, p.segK);
        merge_levels++;
        if (nl == 1) break;
        if (2 * (uint64_t)nl >= n_in) bad_arg("fragment merge would not shrink");
        n_in = 2 * nl;
        if (merge_levels > 64) bad_arg("fragment merge does not terminate");
      }
    }
    uint64_t reduce_levels = 1;
    for (uint32_t chunks = p.T0; chunks > 1; chunks = ceil_div(chunks, 1u << p.logL)) reduce_levels++;
    const uint64_t el = is_g2(curve) ? 2 : 1;
    out[0] = p.c;
    out[1] = p.windows;
    out[2] = p.bucket_windows;
    out[3] = p.entries;
    out[4] = p.K;
    out[5] = p.nlanes;
    out[6] = merge_levels;
    out[7] = reduce_levels;
    out[8] = p.keybits;
    // device bytes of the per-run work buffers (entry buffers x2, grouping scratch, buckets, slots x2, reduce x4)
    out[9] = work_bytes(p, el);
  });
}

RustError mi355_msm_shard_bounds(size_t npoints, int nshards, int shard, size_t* lo, size_t* hi) {
  return guarded([&] {
    if (!lo || !hi) bad_arg("null output");
    if (nshards < 1 || shard < 0 || shard >= nshards) bad_arg("shard %d of %d", shard, nshards);
    shard_bounds(npoints, (size_t)nshards, (size_t)shard, *lo, *hi);
  });
}

#ifdef MSM_DEBUG
const char* mi355_msm_version(void) { return "mi355-msm 0.4 (gfx950) +debug-invariants"; }
#else
const char* mi355_msm_version(void) { return "mi355-msm 0.4 (gfx950)"; }
#endif

}  // extern "C"

#include "msm_stream.hpp"
