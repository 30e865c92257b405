// host_pipeline.hpp -- the host-side concurrency of the engine, free of HIP so that it also compiles into a sanitizer harness
// (tests/tsan_pipeline.cpp: -fsanitize=thread against a fake asynchronous copy engine; tools/sanitize_host.sh).
//
//   UploaderT<Api>   the upload side of a stateless call (msm_stateless.hpp): T staging threads copy the pieces of every slice
//                    into a ring of pinned slots and enqueue one asynchronous copy per piece on the copy stream; one event per
//                    slice; the ring of raw-record device buffers is protected by the consumer's conversion events.
//   run_on_shards    one host thread per shard, first failure re-thrown with the shard named (msm_sharded.hpp).
//
// `Api` supplies the device calls: stream_t, event_t, set_device(int), event_sync(ev), stream_wait(stream, ev),
// copy_h2d(dst, src, bytes, stream), event_record(ev, stream); each throws on failure.
#pragma once

#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace msm_host {

constexpr int RING_SLOTS = 12;
constexpr size_t RING_PIECE = (size_t)16 << 20;

struct PipelineError : std::runtime_error {
  int code;
  PipelineError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

struct Piece {
  const uint8_t* src;
  uint8_t* dst;
  size_t bytes;
  uint32_t slice;
  bool raw;   // lands in the ring of raw-record buffers (bases) rather than in the scalar buffer
};

template <class Api>
struct UploaderT {
  using stream_t = typename Api::stream_t;
  using event_t = typename Api::event_t;
  int device = 0;
  void* const* ring_slot = nullptr;     // RING_SLOTS pinned buffers of RING_PIECE bytes
  const event_t* ring_ev = nullptr;     // the copy out of slot i has completed
  stream_t copy_stream{};
  std::vector<Piece> pieces;
  std::vector<event_t> slice_ev, conv_ev;
  std::unique_ptr<std::atomic<int>[]> slice_left, slice_ready, conv_recorded;
  std::atomic<size_t> next{0};
  std::atomic<size_t> slot_gen[RING_SLOTS];
  std::atomic<int> failed{0};
  std::mutex err_mu;
  std::string err;
  int err_code = 0;
  uint32_t raw_ring = 3;   // device buffers for raw base records: slice s uses buffer s % raw_ring
  std::vector<std::thread> threads;

  // Slice bookkeeping for `slices` slices; pieces are appended by the caller, slice_left[s] = number of pieces of slice s.
  void prepare(uint32_t slices) {
    slice_left.reset(new std::atomic<int>[slices]);
    slice_ready.reset(new std::atomic<int>[slices]);
    conv_recorded.reset(new std::atomic<int>[slices]);
    for (uint32_t s = 0; s < slices; s++) {
      slice_left[s].store(0);
      slice_ready[s].store(0);
      conv_recorded[s].store(0);
    }
    for (auto& g : slot_gen) g.store(0);
  }

  void fail(int code, const std::string& what) {
    std::lock_guard<std::mutex> lk(err_mu);
    if (!failed.exchange(1)) {
      err_code = code ? code : -1;
      err = what;
    }
  }

  void worker() {
    try {
      Api::set_device(device);
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= pieces.size() || failed.load()) return;
        const Piece& pc = pieces[k];
        const int slot = (int)(k % RING_SLOTS);
        const size_t gen = k / RING_SLOTS;
        // the slot's previous piece must have been enqueued (its thread claimed it earlier and depends on nothing later) ...
        while (slot_gen[slot].load(std::memory_order_acquire) != gen) {
          if (failed.load()) return;
          std::this_thread::yield();
        }
        if (gen) Api::event_sync(ring_ev[slot]);   // ... and its copy must have left the slot
        memcpy(ring_slot[slot], pc.src, pc.bytes);
        // the raw-record buffer of slice s held slice s - raw_ring before: that slice's conversion must be ahead of this copy
        if (pc.raw && pc.slice >= raw_ring) {
          const uint32_t dep = pc.slice - raw_ring;
          while (!conv_recorded[dep].load(std::memory_order_acquire)) {
            if (failed.load()) return;
            std::this_thread::yield();
          }
          Api::stream_wait(copy_stream, conv_ev[dep]);
        }
        Api::copy_h2d(pc.dst, ring_slot[slot], pc.bytes, copy_stream);
        Api::event_record(ring_ev[slot], copy_stream);
        slot_gen[slot].store(gen + 1, std::memory_order_release);
        // every copy of the slice is enqueued before its counter reaches zero, so the event covers them all
        if (slice_left[pc.slice].fetch_sub(1) == 1) {
          Api::event_record(slice_ev[pc.slice], copy_stream);
          slice_ready[pc.slice].store(1, std::memory_order_release);
        }
      }
    } catch (const PipelineError& e) {
      fail(e.code, e.what());
    } catch (const std::exception& e) {
      fail(-1, e.what());
    }
  }

  void start(size_t nthreads) {
    for (size_t t = 0; t < nthreads; t++) threads.emplace_back([this] { worker(); });
  }

  // Block until slice s is fully enqueued, then make `st` wait for its copies.
  void await_slice(uint32_t s, stream_t st) {
    while (!slice_ready[s].load(std::memory_order_acquire)) {
      if (failed.load()) throw_failure();
      std::this_thread::yield();
    }
    Api::stream_wait(st, slice_ev[s]);
  }

  // The consumer has enqueued the conversion of slice s and recorded conv_ev[s] behind it.
  void conversion_recorded(uint32_t s) { conv_recorded[s].store(1, std::memory_order_release); }

  [[noreturn]] void throw_failure() {
    std::lock_guard<std::mutex> lk(err_mu);
    throw PipelineError(err_code ? err_code : -1, "stateless upload: " + err);
  }

  void join() {
    for (auto& t : threads)
      if (t.joinable()) t.join();
    threads.clear();
  }

  void abort_and_join() {
    failed.store(1);
    join();
  }
};

// Run fn(g) for g in [0, G) on its own host thread (inline for G = 1); rethrow the first failure as PipelineError with the
// shard named by describe(g).
template <class Fn, class Describe>
void run_on_shards(size_t G, Fn&& fn, Describe&& describe) {
  std::vector<std::string> errors(G);
  std::vector<int> codes(G, 0);
  auto body = [&](size_t g) {
    try {
      fn(g);
    } catch (const PipelineError& e) {
      codes[g] = e.code ? e.code : -1;
      errors[g] = e.what();
    } catch (const std::exception& e) {
      codes[g] = -1;
      errors[g] = e.what();
    } catch (...) {
      codes[g] = -1;
      errors[g] = "unknown C++ exception";
    }
  };
  if (G == 1) {
    body(0);
  } else {
    std::vector<std::thread> threads;
    threads.reserve(G);
    for (size_t g = 0; g < G; g++) threads.emplace_back(body, g);
    for (auto& t : threads) t.join();
  }
  for (size_t g = 0; g < G; g++)
    if (codes[g]) throw PipelineError(codes[g], describe(g) + ": " + errors[g]);
}


// ---- how the operands of a call are cut into pieces (pure arithmetic; tested on the CPU by tests/tsan_pipeline.cpp) -----------------

// The first batch of a host-scalar run -- whose copy nothing can hide -- is handed over in pieces, each computed while the next one
// crosses PCIe (CMB MSM.cu:419-434 splits its first copy 1/4 + 3/4; P1A matter-labs/src/lib.rs:171-182 grows its chunks).
// PCIe delivers 2^26 scalars in ~37 ms and the device works through them in ~110 ms, so a piece can be three times its
// predecessor: n/div, then x 3 each, the last piece taking what is left.  With carried buckets a piece costs one bucket merge
// (~1.5 ms at 2^26), so three pieces pay: n/13, 3n/13, 9n/13 -- the device waits for 2.8 ms of copy instead of 9.4.  Without
// (carry = 0) a piece costs a bucket reduction, a synchronisation and a host fold: 1/4 + 3/4 as before.
// Returns the piece boundaries (front() = 0, back() = n); two entries = no split.
inline std::vector<size_t> first_batch_pieces(size_t n, size_t max_chunk, size_t div) {
  std::vector<size_t> pb{0};
  if (n >= ((size_t)1 << 23) && div >= 2) {
    size_t piece = std::min(n / div, max_chunk);
    while (pb.size() < 7 && piece && pb.back() + piece + piece / 2 < n) {
      pb.push_back(pb.back() + piece);
      piece = std::min(piece * 3, max_chunk);
    }
  }
  pb.push_back(n);
  return pb;
}

// Slice bounds: short first slices so that the first kernels start early (the growing chunks of P1A matter-labs/src/lib.rs:171-182).
// ramp = 1: slice/8, slice/2, then full slices -- the compute side waits for 1/8 slice instead of 1/2 before its first kernel;
// ramp = 0: slice/2, then full slices (the first version; kept for the A/B of profiles/r03_stateless_probe.txt).
// ramp_down (carried buckets only: a further slice then costs one merge, not a bucket reduction): the call ends with slices of
// slice/2, slice/4, slice/8 -- what is left to compute when the last byte has crossed PCIe is an eighth of a slice and the one
// bucket reduction, instead of a whole slice.
inline std::vector<size_t> stateless_slices(size_t n, size_t slice, int ramp, bool ramp_down = false) {
  std::vector<size_t> lo{0};
  if (n == 0) return {0, 0};
  std::vector<size_t> down;
  if (ramp_down && slice >= 64 && n >= 4 * slice) down = {slice / 2, slice / 4, slice / 8};
  size_t tail = 0;
  for (size_t d : down) tail += d;
  const size_t body = n - tail;   // what the ramp-up and the full slices cover
  if (body > slice + slice / 2) {
    if (ramp && slice >= 64) lo.push_back(slice / 8);
    lo.push_back(lo.back() + slice / 2);
  }
  while (lo.back() + slice < body) {
    // do not leave a sliver for the last slice: it would pay a whole bucket reduction (or merge) for a few pairs
    if (body - (lo.back() + slice) < slice / 4) break;
    lo.push_back(lo.back() + slice);
  }
  lo.push_back(body);
  for (size_t d : down) lo.push_back(lo.back() + d);
  return lo;
}

}  // namespace msm_host
