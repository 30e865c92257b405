// msm_stateless.hpp -- the stateless call msm(bases, scalars, n) as a PIPELINE (included by msm_engine.hip).
//
// What it replaces: sppark's mult_pippenger_inf(out, points, npoints, scalars, ffi_affine_sz)
// (SPK poc/blst-cuda/cuda/pippenger_inf.cu:28-35), which hands BOTH operands over in pageable host memory on every call, and
// the way the reference hides that hand-over: sppark uploads the next slice of points and scalars on a second stream while
// the current one is being sorted and accumulated (SPK msm/pippenger.cuh:617-661), yrrid copies the first scalars in quarters
// (CMB MSM.cu:419-434), Matter Labs feeds growing chunks so that the first kernel starts early
// (P1A matter-labs/src/lib.rs:171-182).
//
// Facts this is built on (tools/ubench_h2d.hip, profiles/r03_h2d.txt; 2 x EPYC 9575F, PCIe gen5 x16):
//   * pinned -> HBM: 57 GB/s.  Pageable memory never seen before: 22-27 GB/s through hipMemcpy (the runtime pins the pages in
//     place, ~40 ms/GB), 56 GB/s only on a second pass over the same pages -- which a stateless call does not get.
//   * 4 host threads copying into a ring of pinned pieces sustain 54-56 GB/s on cold pageable memory (1 thread: 30, 2: 44-50).
//   * hipHostMalloc costs ~190 ms/GB: the ring is small (12 x 16 MiB) and kept for the life of the process;
//     hipMalloc / hipFree of tens of GB cost 0.2 ms: device buffers are per call.
//   * round 2's call was serial: one hipMemcpy of 7 GB, conversion, the Edwards image (26 ms + 12.9 GB for a law that saves
//     18 ms on a single run), a second copy for the scalars, then the MSM -- 307 ms at best, 1003 ms on the driver's box.
//
// The pipeline: the pairs are cut into slices (2^22 first, then 2^23).  T staging threads copy the bases and the scalars of
// slice s piece by piece into the pinned ring and enqueue one DMA per piece on the copy stream; the compute stream waits for
// the slice's event, converts its bases (104-B arkworks records -> radix-2^28 limbs; raw records live in a ring of three
// device buffers) and runs the ordinary single-chunk pipeline on it (digits, grouping, accumulation on the XYZZ law, bucket
// reduction, host fold) while slices s+1, s+2 cross PCIe.  The partial sums are added on the host, like chunks of a
// context run -- the result is the same normalised point.  At 2^26 the call is PCIe-bound: 9.1 GB / 56 GB/s = 163 ms plus
// the last slice's compute.
#pragma once

#include <atomic>
#include <memory>
#include <mutex>

namespace {

struct StageRing {
  static constexpr int SLOTS = msm_host::RING_SLOTS;
  static constexpr size_t PIECE = msm_host::RING_PIECE;
  int device = -1;
  void* slot[SLOTS] = {};
  hipEvent_t ev[SLOTS] = {};
};

std::mutex g_ring_mu;
std::vector<StageRing*> g_rings_idle;   // kept for the life of the process (mi355_msm_trim gives them back)

void ring_destroy(StageRing* r) {
  if (!r) return;
  for (int i = 0; i < StageRing::SLOTS; i++) {
    if (r->slot[i]) (void)hipHostFree(r->slot[i]);
    if (r->ev[i]) (void)hipEventDestroy(r->ev[i]);
  }
  delete r;
}

// A ring for `device` (the calling thread's current device): an idle one if there is one, a new one otherwise.
StageRing* ring_acquire(int device) {
  {
    std::lock_guard<std::mutex> lk(g_ring_mu);
    for (size_t i = 0; i < g_rings_idle.size(); i++)
      if (g_rings_idle[i]->device == device) {
        StageRing* r = g_rings_idle[i];
        g_rings_idle.erase(g_rings_idle.begin() + (long)i);
        return r;
      }
  }
  StageRing* r = new StageRing();
  r->device = device;
  try {
    for (int i = 0; i < StageRing::SLOTS; i++) {
      HIP_OK(hipHostMalloc(&r->slot[i], StageRing::PIECE, hipHostMallocDefault));
      HIP_OK(hipEventCreateWithFlags(&r->ev[i], hipEventDisableTiming));
    }
  } catch (...) {
    ring_destroy(r);
    throw;
  }
  return r;
}

void ring_release(StageRing* r) {
  std::lock_guard<std::mutex> lk(g_ring_mu);
  g_rings_idle.push_back(r);
}

// The device side of a stateless call -- a context with its raw-record, base, scalar and work buffers (~9 GB at 2^26) -- is
// kept between calls too: measured, a call that frees its buffers makes the NEXT call's first hipMalloc wait ~280 ms
// (the driver releases the memory asynchronously; profiles/r03_stateless_probe.txt), and growing the work buffers from the
// short first slice to the regular one cost 30 ms inside the pipeline.
//
// What is retained, and its bounds: at most ONE idle context per (curve, device) -- a second concurrent caller's context
// is destroyed when it comes back and finds the slot taken; a context that holds more than MI355_MSM_STATELESS_KEEP_MB of device
// memory (default 16384: a 2^26-pair G1 call holds ~9 GB) gives its buffers back before it is parked (the next call of that size
// pays the allocation again); any device allocation of this library that fails with out-of-memory first frees ALL idle contexts and
// retries (DevBuf::reserve -> reclaim_idle_device_memory) before a run falls back to smaller chunks; mi355_msm_trim() frees them
// and the pinned rings on request.  Documented in INTEGRATION.md section 4.
std::vector<mi355_msm_ctx*> g_stateless_idle;

long env_long(const char* name, long dflt, long lo, long hi);

struct StatelessLease {
  mi355_msm_ctx* ctx = nullptr;
  bool ok = false;
  StatelessLease(int curve, int device) {
    if (device < 0) HIP_OK(hipGetDevice(&device));
    {
      std::lock_guard<std::mutex> lk(g_ring_mu);
      for (size_t i = 0; i < g_stateless_idle.size(); i++)
        if (g_stateless_idle[i]->curve == curve && g_stateless_idle[i]->device == device) {
          ctx = g_stateless_idle[i];
          g_stateless_idle.erase(g_stateless_idle.begin() + (long)i);
          return;
        }
    }
    take(mi355_msm_create(&ctx, curve, device));
    ctx->opt_anchor = 0;   // new bases with every call: the sum of the bases an anchored window needs would be computed every time
  }
  void keep() { ok = true; }   // the call went through: the context goes back to the pool instead of being destroyed
  ~StatelessLease() {
    if (!ctx) return;
    if (ok) {
      const long keep_mb = env_long("MI355_MSM_STATELESS_KEEP_MB", 16384, 0, 1l << 30);
      if (ctx_device_bytes(ctx) > ((size_t)keep_mb << 20)) {
        release_work_buffers(ctx);
        ctx->scalars.release();
        ctx->bases.release();
        ctx->inf.release();
        for (DevBuf& r : ctx->stateless_raw) r.release();
      }
      std::vector<mi355_msm_ctx*> evict;
      {
        std::lock_guard<std::mutex> lk(g_ring_mu);
        bool taken = false;
        for (mi355_msm_ctx* c : g_stateless_idle) taken = taken || (c->curve == ctx->curve && c->device == ctx->device);
        if (!taken) {
          // the bound is on what sits idle on the DEVICE, all curves together (four curves x 16 GB could sit there while
          // another allocator of the process -- torch, RCCL -- runs out): the contexts parked longest go first
          size_t idle = ctx_device_bytes(ctx);
          for (mi355_msm_ctx* c : g_stateless_idle)
            if (c->device == ctx->device) idle += ctx_device_bytes(c);
          for (size_t i = 0; i < g_stateless_idle.size() && idle > ((size_t)keep_mb << 20);) {
            if (g_stateless_idle[i]->device != ctx->device) { i++; continue; }
            idle -= ctx_device_bytes(g_stateless_idle[i]);
            evict.push_back(g_stateless_idle[i]);
            g_stateless_idle.erase(g_stateless_idle.begin() + (long)i);
          }
          g_stateless_idle.push_back(ctx);
          ctx = nullptr;
        }
      }
      for (mi355_msm_ctx* c : evict) {
        RustError d = mi355_msm_destroy(c);
        if (d.message) free(d.message);
      }
    }
    if (ctx) {   // a failed call, or the slot of this (curve, device) is taken: one idle context per key, no more
      RustError d = mi355_msm_destroy(ctx);
      if (d.message) free(d.message);
    }
  }
  StatelessLease(const StatelessLease&) = delete;
  StatelessLease& operator=(const StatelessLease&) = delete;
};

bool reclaim_idle_device_memory() {
  std::vector<mi355_msm_ctx*> idle;
  {
    std::lock_guard<std::mutex> lk(g_ring_mu);
    idle.swap(g_stateless_idle);
  }
  if (idle.empty()) return false;
  DeviceGuard keep;   // destroying a context of another device must not move the caller
  for (mi355_msm_ctx* c : idle) {
    RustError d = mi355_msm_destroy(c);
    if (d.message) free(d.message);
  }
  return true;
}

// What the most recent stateless call of this thread did (mi355_msm_last_stateless).
struct StatelessStats {
  double total_ms = 0, setup_ms = 0, wait_upload_ms = 0, compute_ms = 0, tail_ms = 0;
  double slices = 0, threads = 0, bytes = 0;
  double dma_done_ms = 0;       // when the LAST DMA of the call completed, from the start of the call (device clock of the copy stream)
  double first_dma_ms = 0;      // when the copy stream started
};
thread_local StatelessStats g_last_stateless;

long env_long(const char* name, long dflt, long lo, long hi) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  char* end = nullptr;
  const long v = strtol(e, &end, 10);
  if (*end || v < lo || v > hi) return dflt;
  return v;
}

// The device calls of the staging pipeline (host_pipeline.hpp UploaderT<Api>).
struct HipPipelineApi {
  using stream_t = hipStream_t;
  using event_t = hipEvent_t;
  static void set_device(int d) { HIP_OK(hipSetDevice(d)); }
  static void event_sync(hipEvent_t e) { HIP_OK(hipEventSynchronize(e)); }
  static void stream_wait(hipStream_t s, hipEvent_t e) { HIP_OK(hipStreamWaitEvent(s, e, 0)); }
  static void copy_h2d(void* dst, const void* src, size_t bytes, hipStream_t s) { HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s)); }
  static void event_record(hipEvent_t e, hipStream_t s) { HIP_OK(hipEventRecord(e, s)); }
};
using msm_host::Piece;

// The upload side of one stateless call: the staging threads of UploaderT plus the HIP objects it drives.
struct Uploader : msm_host::UploaderT<HipPipelineApi> {
  ~Uploader() {
    abort_and_join();   // a compute-side failure: let the staging threads drain
    if (copy_stream) {
      (void)hipStreamSynchronize(copy_stream);   // copies out of the ring may still be in flight: the ring goes back to the pool next
      (void)hipStreamDestroy(copy_stream);
    }
    for (auto e : slice_ev)
      if (e) (void)hipEventDestroy(e);
    for (auto e : conv_ev)
      if (e) (void)hipEventDestroy(e);
  }
};

template <class C>
void stateless_t(mi355_msm_ctx* ctx, uint8_t* out, const uint8_t* affine, size_t n, const uint8_t* scalars, size_t stride) {
  using E = typename C::E;
  using AD = AffineDevT<typename E::T>;
  const auto t_begin = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  };
  hipStream_t st = ctx->own_stream;
  StatelessStats stats;
  typename HostTail<E>::Pt total;
  HostTail<E>::set_inf(total);
  memset(ctx->last_ms, 0, sizeof ctx->last_ms);
  memset(ctx->last_info, 0, sizeof ctx->last_info);
  ctx->pre_c = ctx->pre_windows = 0;
  ctx->te_active = false;
  ctx->sw_level0_only = false;
  ctx->fitted_chunk = 0;
  if (n) {
    // slices of n/4 .. n/8 pairs, between 2^20 (below that a slice's fixed costs -- one bucket reduction each -- outweigh the
    // overlap) and 2^23 (1.1 GB: 20 ms of PCIe against ~20 ms of compute)
    const long auto_log = std::min<long>(23, std::max<long>(20, (long)ilog2_floor(std::max<size_t>(n, 4) / 4)));
    const size_t slice = (size_t)1 << env_long("MI355_MSM_STATELESS_SLICE_LOG", auto_log, 10, 26);
    const bool want_carry = ctx->opt_carry && env_long("MI355_MSM_STATELESS_CARRY", 0, 0, 1);
    const std::vector<size_t> lo = msm_host::stateless_slices(n, slice, (int)env_long("MI355_MSM_STATELESS_RAMP", 1, 0, 1),
                                                    env_long("MI355_MSM_STATELESS_RAMP_DOWN", 0, 0, 1) != 0);
    const uint32_t S = (uint32_t)lo.size() - 1;
    size_t max_cnt = 0;
    for (uint32_t s = 0; s < S; s++) max_cnt = std::max(max_cnt, lo[s + 1] - lo[s]);

    // the slices share one bucket array (BucketCarry, msm_engine.hip): every slice groups and accumulates with the window size of
    // the whole call, only the last one reduces; 0 = every slice reduces its own buckets with its own window size
    const long forced_c = env_long("MI355_MSM_STATELESS_CARRY_C", 0, 0, 23);   // A/B only: the carried slices' window size
    const uint32_t carry_c = (S > 1 && want_carry) ? (forced_c ? (uint32_t)forced_c : ctx->plan(std::min(n, (size_t)1 << 26), false).c) : 0;
    BucketCarry carry{carry_c, 0, true, false};
    const bool trace = env_long("MI355_MSM_STATELESS_TRACE", 0, 0, 1) != 0;   // per-slice timeline on stderr
    std::vector<double> tr_ready(S), tr_done(S);
    std::vector<std::pair<const char*, double>> tr_setup;
    hipEvent_t tr_t0 = nullptr;
    Uploader up;
    up.device = ctx->device;
    up.copy_stream = nullptr;
    up.raw_ring = std::min<uint32_t>(3, S);
    DevBuf* raw = ctx->stateless_raw;
    struct RingLease {
      StageRing* r = nullptr;
      ~RingLease() { if (r) ring_release(r); }
    } lease;
    try {
      for (uint32_t i = 0; i < up.raw_ring; i++) raw[i].reserve(max_cnt * stride);
      tr_setup.emplace_back("raw buffers", ms_since(t_begin));
      ctx->bases.reserve(max_cnt * sizeof(AD));
      ctx->inf.reserve(max_cnt);
      ctx->scalars.reserve(n * 32);
      {
        // the work buffers of every slice size at once (a short first slice would otherwise be followed by a free + malloc)
        WorkBytes w;
        bool fits = true;
        for (uint32_t s = 0; s < S; s++) {
          const size_t cnt = lo[s + 1] - lo[s];
          if (s > 2 && s + 4 < S) continue;   // the slices in between have the size of slice 2
          const Plan p = ctx->plan(cnt, false, carry_c);
          fits = fits && p.entries < (1ull << 32);
          w.max_with(chunk_work_bytes(p, cnt, false, sizeof(XyzzDevT<typename E::T>), carry_c != 0));
        }
        if (fits && !ctx->opt_mem_limit) {
          try {
            reserve_work(ctx, w);
          } catch (const HipFailure& e) {
            if (e.code != (int)hipErrorOutOfMemory) throw;
            release_work_buffers(ctx);   // short of memory: let the per-chunk fit / back-off below size the chunks
          }
        }
      }
      tr_setup.emplace_back("bases + scalars + work buffers", ms_since(t_begin));
      lease.r = ring_acquire(ctx->device);
      up.ring_slot = lease.r->slot;
      up.ring_ev = lease.r->ev;
      tr_setup.emplace_back("ring", ms_since(t_begin));
      up.copy_stream = create_copy_stream();   // high priority: its own hardware queue, never behind the compute stream's kernels
      up.slice_ev.assign(S, nullptr);
      up.conv_ev.assign(S, nullptr);
      for (uint32_t s = 0; s < S; s++) {
        // (the last slice's event is always a timing one: "when were the uploads done" is part of the call's report)
        HIP_OK(hipEventCreateWithFlags(&up.slice_ev[s], (trace || s + 1 == S) ? hipEventDefault : hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&up.conv_ev[s], trace ? hipEventDefault : hipEventDisableTiming));
      }
      HIP_OK(hipEventCreate(&tr_t0));
      HIP_OK(hipEventRecord(tr_t0, up.copy_stream));
      stats.first_dma_ms = ms_since(t_begin);
      tr_setup.emplace_back("stream + events", ms_since(t_begin));
      up.prepare(S);
      // pieces in upload order: the scalars of a slice first (its grouping needs them before the accumulation needs bases)
      for (uint32_t s = 0; s < S; s++) {
        const size_t cnt = lo[s + 1] - lo[s];
        int count = 0;
        auto cut = [&](const uint8_t* src, uint8_t* dst, size_t bytes, bool is_raw) {
          for (size_t off = 0; off < bytes; off += StageRing::PIECE) {
            up.pieces.push_back(Piece{src + off, dst + off, std::min(StageRing::PIECE, bytes - off), s, is_raw});
            count++;
          }
        };
        cut(scalars + lo[s] * 32, ctx->scalars.as<uint8_t>() + lo[s] * 32, cnt * 32, false);
        cut(affine + lo[s] * stride, raw[s % up.raw_ring].as<uint8_t>(), cnt * stride, true);
        up.slice_left[s].store(count);
      }
      const long want = env_long("MI355_MSM_STAGE_THREADS", 6, 1, StageRing::SLOTS);
      const size_t T = std::min<size_t>((size_t)want, up.pieces.size());
      stats.threads = (double)T;
      stats.slices = S;
      stats.bytes = (double)(n * (stride + 32));
      up.start(T);
      stats.setup_ms = ms_since(t_begin);

      for (uint32_t s = 0; s < S; s++) {
        const size_t cnt = lo[s + 1] - lo[s];
        const auto t_wait = std::chrono::steady_clock::now();
        up.await_slice(s, st);
        stats.wait_upload_ms += ms_since(t_wait);
        tr_ready[s] = ms_since(t_begin);
        const auto t_comp = std::chrono::steady_clock::now();
        HIP_OK(Launch<E>::convert_bases(raw[s % up.raw_ring].as<uint8_t>(), stride, (uint32_t)cnt, false, ctx->bases.as<AD>(), ctx->inf.as<uint8_t>(), st));
        HIP_OK(hipEventRecord(up.conv_ev[s], st));
        up.conversion_recorded(s);
        ctx->nbases = cnt;
        // the slice as chunks of the ordinary pipeline (one, unless device memory is short)
        size_t max_chunk = ctx->opt_max_chunk ? (size_t)ctx->opt_max_chunk : ((size_t)1 << 26);
        size_t reclaimed_at = (size_t)-1;
        for (size_t off = 0; off < cnt;) {
          size_t cn = std::min(max_chunk, cnt - off);
          if (cn > ctx->fitted_chunk) cn = fit_chunk(ctx, cn, false, carry_c);
          typename HostTail<E>::Pt part;
          carry.last = s + 1 == S && off + cn >= cnt;
          try {
            run_chunk<C>(ctx, ctx->scalars.as<uint32_t>() + (lo[s] + off) * 8, off, cn, st, part, nullptr, carry_c ? &carry : nullptr);
          } catch (const HipFailure& e) {
            if (e.code != (int)hipErrorOutOfMemory || cn <= 1024) throw;
            (void)hipStreamSynchronize(st);
            release_work_buffers(ctx, carry_c && !carry.first);   // (the totals of the slices so far stay)
            if (reclaimed_at != off && reclaim_idle_device_memory()) {   // parked contexts of other curves / devices go first, once
              reclaimed_at = off;
              continue;
            }
            max_chunk = ctx->chunk_cap = (cn + 1) / 2;
            ctx->oom_backoffs++;
            continue;
          }
          if (cn > ctx->fitted_chunk) ctx->fitted_chunk = cn;
          HostTail<E>::add(total, part);
          off += cn;
          carry.first = false;
          carry.index++;
        }
        const double c_ms = ms_since(t_comp);
        tr_done[s] = ms_since(t_begin);
        stats.compute_ms += c_ms;
        if (s + 1 == S) stats.tail_ms = c_ms;
        if (up.failed.load()) up.throw_failure();
      }
      up.join();
      if (up.failed.load() && !up.err.empty()) up.throw_failure();
      if (trace) {
        fprintf(stderr, "[mi355_msm stateless] n=%zu slices=%u threads=%d setup:", n, S, (int)stats.threads);
        for (auto& kv : tr_setup) fprintf(stderr, "  %s @%.1f", kv.first, kv.second);
        fprintf(stderr, "  threads started @%.1f ms\n", stats.setup_ms);
        for (uint32_t s = 0; s < S; s++) {
          float dma = 0, conv = 0;
          (void)hipEventElapsedTime(&dma, tr_t0, up.slice_ev[s]);
          (void)hipEventElapsedTime(&conv, tr_t0, up.conv_ev[s]);
          fprintf(stderr, "  slice %2u  pairs %9zu  all DMAs enqueued @%7.1f (host)  DMA done +%7.1f  converted +%7.1f (device, from the copy stream's start)  chunk folded @%7.1f (host)\n",
                  s, lo[s + 1] - lo[s], tr_ready[s], dma, conv, tr_done[s]);
        }
      }
      {
        float dma = 0;
        if (hipEventElapsedTime(&dma, tr_t0, up.slice_ev[S - 1]) == hipSuccess) stats.dma_done_ms = stats.first_dma_ms + dma;
        (void)hipGetLastError();
        (void)hipEventDestroy(tr_t0);
        tr_t0 = nullptr;
      }
    } catch (...) {
      if (tr_t0) (void)hipEventDestroy(tr_t0);
      up.abort_and_join();
      (void)hipStreamSynchronize(st);
      if (up.copy_stream) (void)hipStreamSynchronize(up.copy_stream);
      throw;
    }
    (void)hipStreamSynchronize(up.copy_stream);
  }
  HostTail<E>::to_abi(out, total);
  stats.total_ms = ms_since(t_begin);
  g_last_stateless = stats;
}

void stateless_run(mi355_msm_ctx* ctx, void* out, const void* affine, size_t n, const void* scalars, size_t stride) {
  ensure_device(ctx);
  const size_t min_stride = 2 * coord_bytes(ctx->curve) + 1;
  if (stride < min_stride || (stride & 3)) bad_arg("affine stride %zu is not a 4-byte multiple >= %zu", stride, min_stride);
  if (n >= (1ull << 31)) bad_arg("npoints %zu exceeds 2^31-1", n);
  if (!out) bad_arg("null output pointer");
  if (n && (!affine || !scalars)) bad_arg("null bases or scalars pointer");
  with_curve(ctx->curve, [&]<class C>() { stateless_t<C>(ctx, (uint8_t*)out, (const uint8_t*)affine, n, (const uint8_t*)scalars, stride); });
}

}  // namespace
