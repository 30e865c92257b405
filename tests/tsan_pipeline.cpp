// tsan_pipeline.cpp -- the engine's host concurrency (csrc/host_pipeline.hpp) under ThreadSanitizer, without a GPU.
//
//   g++ -O1 -g -std=c++20 -fsanitize=thread -pthread -I2022-entries_amd/csrc -o /tmp/tsan_pipeline tests/tsan_pipeline.cpp && /tmp/tsan_pipeline
//   (tools/sanitize_host.sh runs it under -fsanitize=thread and under -fsanitize=address,undefined; profiles/r03_sanitizers.txt)
//
// The device is replaced by a FAKE ASYNCHRONOUS COPY ENGINE with HIP's ordering rules: a stream is an in-order command queue run
// by its own thread; an event completes when the queue reaches its record; stream_wait blocks the queue until the event's most
// recent record has completed.  Copies really happen later than they are enqueued, so a staging thread that reused a ring slot
// before its copy had left it, a copy into a raw-record buffer that the consumer had not converted yet, or a slice event that
// did not cover all of the slice's copies would be a data race ThreadSanitizer reports AND a checksum mismatch counted here.
// The consumer mirrors stateless_t (msm_stateless.hpp): await_slice, "convert" the slice out of its raw buffer on the compute
// stream, record the conversion event, go on.  Then: eight pipelines at once through run_on_shards (one ring, two streams and
// six staging threads each), and the failure paths (a copy that throws; a shard that throws).
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>

#include "host_pipeline.hpp"

using namespace msm_host;

struct FakeEvent {
  std::atomic<uint64_t> issued{0}, completed{0};
};

struct FakeStream {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false;
  std::thread th;
  FakeStream() : th([this] { run(); }) {}
  ~FakeStream() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    th.join();
  }
  void run() {
    for (;;) {
      std::function<void()> cmd;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return stop || !q.empty(); });
        if (q.empty()) return;
        cmd = std::move(q.front());
        q.pop_front();
      }
      cmd();
    }
  }
  void enqueue(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lk(mu);
      q.push_back(std::move(f));
    }
    cv.notify_one();
  }
  void sync() {
    FakeEvent e;
    const uint64_t seq = ++e.issued;
    enqueue([&e, seq] { e.completed.store(seq, std::memory_order_release); });
    while (e.completed.load(std::memory_order_acquire) < seq) std::this_thread::yield();
  }
};

static std::atomic<long> g_fail_copy_after{-1};   // > 0: the n-th copy from now on throws (failure-path test)

struct FakeApi {
  using stream_t = FakeStream*;
  using event_t = FakeEvent*;
  static void set_device(int) {}
  static void event_sync(FakeEvent* e) {
    const uint64_t target = e->issued.load(std::memory_order_acquire);
    while (e->completed.load(std::memory_order_acquire) < target) std::this_thread::yield();
  }
  static void stream_wait(FakeStream* s, FakeEvent* e) {
    const uint64_t target = e->issued.load(std::memory_order_acquire);
    s->enqueue([e, target] {
      while (e->completed.load(std::memory_order_acquire) < target) std::this_thread::yield();
    });
  }
  static void copy_h2d(void* dst, const void* src, size_t bytes, FakeStream* s) {
    if (g_fail_copy_after.load() > 0 && g_fail_copy_after.fetch_sub(1) == 1) throw PipelineError(719, "injected copy failure");
    s->enqueue([=] { memcpy(dst, src, bytes); });
  }
  static void event_record(FakeEvent* e, FakeStream* s) {
    // (HIP: a later record supersedes an earlier one; the staging threads re-record a slot's event only after syncing on it)
    const uint64_t seq = e->issued.fetch_add(1, std::memory_order_acq_rel) + 1;
    s->enqueue([e, seq] {
      uint64_t cur = e->completed.load(std::memory_order_relaxed);
      while (cur < seq && !e->completed.compare_exchange_weak(cur, seq, std::memory_order_release)) {
      }
    });
  }
};

static uint64_t mix(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  return x;
}

// One pipeline run: `slices` slices of `pairs` pairs (stride-byte "bases", 32-byte "scalars"), pieces of `piece` bytes.
// Returns the number of mismatching bytes the consumer saw (0 = every slice arrived whole, in its own raw buffer, in time).
static size_t run_pipeline(uint32_t slices, size_t pairs, size_t stride, size_t piece, size_t threads, uint64_t seed, bool expect_failure = false) {
  const size_t n = slices * pairs;
  std::vector<uint8_t> bases(n * stride), scalars(n * 32);
  for (size_t i = 0; i < bases.size(); i += 8) {
    const uint64_t v = mix(seed + i);
    memcpy(bases.data() + i, &v, std::min<size_t>(8, bases.size() - i));
  }
  for (size_t i = 0; i < scalars.size(); i += 8) {
    const uint64_t v = mix(~seed + i);
    memcpy(scalars.data() + i, &v, 8);
  }
  std::vector<std::vector<uint8_t>> ring(RING_SLOTS, std::vector<uint8_t>(piece));
  std::vector<void*> ring_ptr(RING_SLOTS);
  std::vector<FakeEvent> ring_ev_store(RING_SLOTS);
  std::vector<FakeEvent*> ring_ev(RING_SLOTS);
  for (int i = 0; i < RING_SLOTS; i++) {
    ring_ptr[i] = ring[i].data();
    ring_ev[i] = &ring_ev_store[i];
  }
  const uint32_t R = std::min<uint32_t>(3, slices);
  std::vector<std::vector<uint8_t>> raw(R, std::vector<uint8_t>(pairs * stride));   // "device" raw-record buffers
  std::vector<uint8_t> dev_scalars(n * 32);
  std::vector<FakeEvent> slice_ev(slices), conv_ev(slices);
  FakeStream copy_stream, compute_stream;
  std::atomic<size_t> mismatches{0};
  size_t result = 0;
  {
    UploaderT<FakeApi> up;
    up.ring_slot = ring_ptr.data();
    up.ring_ev = ring_ev.data();
    up.copy_stream = &copy_stream;
    up.raw_ring = R;
    up.prepare(slices);
    for (uint32_t s = 0; s < slices; s++) {
      up.slice_ev.push_back(&slice_ev[s]);
      up.conv_ev.push_back(&conv_ev[s]);
      int count = 0;
      auto cut = [&](const uint8_t* src, uint8_t* dst, size_t bytes, bool is_raw) {
        for (size_t off = 0; off < bytes; off += piece) {
          up.pieces.push_back(Piece{src + off, dst + off, std::min(piece, bytes - off), s, is_raw});
          count++;
        }
      };
      cut(scalars.data() + s * pairs * 32, dev_scalars.data() + s * pairs * 32, pairs * 32, false);
      cut(bases.data() + s * pairs * stride, raw[s % R].data(), pairs * stride, true);
      up.slice_left[s].store(count);
    }
    up.start(threads);
    try {
      for (uint32_t s = 0; s < slices; s++) {
        up.await_slice(s, &compute_stream);
        // "conversion": the slice must be whole in ITS raw buffer, and its scalars in place, when the compute stream gets here
        compute_stream.enqueue([&, s] {
          size_t bad = 0;
          const uint8_t* want = bases.data() + s * pairs * stride;
          const uint8_t* got = raw[s % R].data();
          for (size_t i = 0; i < pairs * stride; i++) bad += want[i] != got[i];
          for (size_t i = 0; i < pairs * 32; i++) bad += scalars[s * pairs * 32 + i] != dev_scalars[s * pairs * 32 + i];
          mismatches.fetch_add(bad);
        });
        FakeApi::event_record(&conv_ev[s], &compute_stream);
        up.conversion_recorded(s);
        if (up.failed.load()) up.throw_failure();
      }
      up.join();
      if (up.failed.load()) up.throw_failure();
      if (expect_failure) {
        fprintf(stderr, "expected a failure, the pipeline went through\n");
        exit(2);
      }
    } catch (const PipelineError& e) {
      up.abort_and_join();
      if (!expect_failure) {
        fprintf(stderr, "unexpected failure: %s\n", e.what());
        exit(2);
      }
      if (e.code != 719 || std::string(e.what()).find("injected copy failure") == std::string::npos) {
        fprintf(stderr, "wrong failure: %d %s\n", e.code, e.what());
        exit(2);
      }
      result = (size_t)-1;
    }
    copy_stream.sync();
    compute_stream.sync();
  }
  return result ? result : mismatches.load();
}

// The cuts of a call's operands (host_pipeline.hpp): every pair belongs to exactly one piece, pieces grow as documented.
static bool check_cuts() {
  using msm_host::first_batch_pieces;
  using msm_host::stateless_slices;
  auto partition_of = [](const std::vector<size_t>& b, size_t n) {
    if (b.size() < 2 || b.front() != 0 || b.back() != n) return false;
    for (size_t i = 1; i < b.size(); i++)
      if (b[i] < b[i - 1] || (b[i] == b[i - 1] && n != 0)) return false;
    return true;
  };
  const size_t N = (size_t)1 << 26, MC = (size_t)1 << 26;
  // first host-scalar batch: n/div, then x 3, the last piece takes the rest; no split below 2^23 pairs
  if (first_batch_pieces(N, MC, 13) != std::vector<size_t>{0, N / 13, N / 13 + 3 * (N / 13), N}) return false;
  if (first_batch_pieces(N, MC, 4) != std::vector<size_t>{0, N / 4, N}) return false;
  if (first_batch_pieces(N, MC, 40).size() != 5) return false;
  if (first_batch_pieces(((size_t)1 << 23) - 1, MC, 13) != std::vector<size_t>{0, ((size_t)1 << 23) - 1}) return false;
  if (first_batch_pieces(0, MC, 13) != std::vector<size_t>{0, 0}) return false;
  for (size_t n : {(size_t)1 << 23, ((size_t)1 << 23) + 7, (size_t)3 << 24, N + 12345, (size_t)1 << 28})
    for (size_t div : {2, 4, 8, 13, 16, 40, 64})
      for (size_t mc : {(size_t)1 << 20, (size_t)1 << 26}) {
        const std::vector<size_t> b = first_batch_pieces(n, mc, div);
        if (!partition_of(b, n) || b.size() > 8) return false;
        for (size_t i = 1; i + 1 < b.size(); i++)
          if (b[i] - b[i - 1] > mc) return false;   // every piece but the last fits a chunk
      }
  // stateless slices: a 1/8 + 1/2 ramp, full slices, optionally a 1/2 + 1/4 + 1/8 ramp-down; no slivers
  const size_t S = (size_t)1 << 23;
  if (stateless_slices(0, S, 1) != std::vector<size_t>{0, 0}) return false;
  if (stateless_slices(S, S, 1) != std::vector<size_t>{0, S}) return false;
  const std::vector<size_t> a = stateless_slices(N, S, 1);
  if (!partition_of(a, N) || a[1] != S / 8 || a[2] != S / 8 + S / 2 || a.size() != 11) return false;
  const std::vector<size_t> d = stateless_slices(N, S, 1, true);
  if (!partition_of(d, N) || d[d.size() - 1] - d[d.size() - 2] != S / 8 || d[d.size() - 2] - d[d.size() - 3] != S / 4) return false;
  for (size_t n : {(size_t)1, (size_t)1000, S - 1, S + 1, 3 * S / 2 + 1, 5 * S + 17, N + 999})
    for (size_t slice : {(size_t)64, (size_t)1 << 20, S})
      for (int ramp : {0, 1})
        for (bool down : {false, true}) {
          const std::vector<size_t> b = stateless_slices(n, slice, ramp, down);
          if (!partition_of(b, n)) return false;
          for (size_t i = 1; i < b.size(); i++)
            if (b[i] - b[i - 1] > slice + slice / 2 + slice / 4) return false;   // the last full slice may absorb a remainder below slice/4 ... of the body
        }
  return true;
}

int main() {
  if (!check_cuts()) {
    fprintf(stderr, "piece / slice boundaries are wrong\n");
    return 1;
  }
  printf("piece and slice boundaries: partitions of the input, documented shapes\n");
  // many small pieces: the ring of 12 slots wraps dozens of times, the three raw buffers a few times
  struct Case {
    uint32_t slices;
    size_t pairs, stride, piece, threads;
  } cases[] = {{1, 1000, 104, 4096, 1}, {2, 3001, 104, 8192, 2}, {7, 20000, 104, 32768, 6}, {9, 9999, 200, 16384, 12}, {5, 50000, 112, 65536, 4}};
  int idx = 0;
  for (const Case& c : cases) {
    const size_t bad = run_pipeline(c.slices, c.pairs, c.stride, c.piece, c.threads, 1000 + idx);
    printf("pipeline %d: %u slices x %zu pairs, %zu-byte pieces, %zu staging threads: %zu mismatching bytes\n", idx, c.slices, c.pairs, c.piece, c.threads, bad);
    if (bad) return 1;
    idx++;
  }
  // eight pipelines at once, one per "shard", through the shard fan-out
  std::vector<size_t> bad(8, 1);
  run_on_shards(8, [&](size_t g) { bad[g] = run_pipeline(4, 6000 + 100 * g, 104, 16384, 6, 77 + g); }, [](size_t g) { return "shard " + std::to_string(g); });
  size_t total = 0;
  for (size_t b : bad) total += b;
  printf("8 concurrent pipelines through run_on_shards: %zu mismatching bytes\n", total);
  if (total) return 1;
  // failure paths: a copy that throws inside a staging thread reaches the consumer with its code and message ...
  g_fail_copy_after.store(17);
  if (run_pipeline(6, 8000, 104, 16384, 6, 5, true) != (size_t)-1) return 1;
  g_fail_copy_after.store(-1);
  printf("injected copy failure: reported to the consumer, threads drained\n");
  // ... and a shard that throws is named in the error the caller sees, after every other shard has finished
  std::atomic<int> finished{0};
  try {
    run_on_shards(5, [&](size_t g) {
      if (g == 3) throw PipelineError(2, "out of memory");
      run_pipeline(2, 2000, 104, 8192, 3, g);
      finished++;
    }, [](size_t g) { return "shard " + std::to_string(g) + " (device " + std::to_string(g) + ")"; });
    return 1;
  } catch (const PipelineError& e) {
    if (e.code != 2 || std::string(e.what()) != "shard 3 (device 3): out of memory" || finished.load() != 4) {
      fprintf(stderr, "wrong shard failure: %d '%s' finished=%d\n", e.code, e.what(), finished.load());
      return 1;
    }
  }
  printf("shard failure: named, other shards completed\n");
  printf("OK\n");
  return 0;
}
