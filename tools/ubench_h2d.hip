// ubench_h2d.hip -- how fast can PAGEABLE host memory reach HBM on this box?  (facts behind the stateless mi355_msm() pipeline)
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_h2d tools/ubench_h2d.hip -lpthread
//
// Variants, all moving the same `total` bytes out of an ordinary malloc'ed (touched) buffer:
//   A  one synchronous hipMemcpy (what set_bases_host did in round 2)
//   B  the same bytes from PINNED memory (the PCIe ceiling)
//   C  T host threads memcpy slices into a ring of pinned buffers, one DMA per slice behind them (T = 1, 2, 4, 8, 16)
//   D  hipHostRegister on slices of the caller's buffer, DMA straight out of it, unregister (page pinning in place)
//   E  the first-touch cost of a fresh pinned allocation of the ring (what a cold first call pays)
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define OK(e)                                                                      \
  do {                                                                             \
    hipError_t e_ = (e);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(e_), __LINE__);  \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const size_t total = (argc > 1 ? (size_t)atol(argv[1]) : 4096) << 20;
  const size_t slice = (argc > 2 ? (size_t)atol(argv[2]) : 64) << 20;
  printf("host threads available: %u, total %zu MiB, slice %zu MiB\n", std::thread::hardware_concurrency(), total >> 20, slice >> 20);
  uint8_t* src = (uint8_t*)malloc(total);
  {
    // touch with several threads (first touch decides the NUMA node of a page)
    std::vector<std::thread> th;
    for (int t = 0; t < 8; t++)
      th.emplace_back([&, t] { memset(src + total / 8 * t, t + 1, total / 8); });
    for (auto& x : th) x.join();
  }
  void* dst = nullptr;
  OK(hipMalloc(&dst, total));
  hipStream_t st;
  OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    OK(hipMemcpy(dst, src, total, hipMemcpyHostToDevice));
    double dt = now() - t0;
    printf("A  hipMemcpy from pageable (rep %d):            %7.1f ms  %6.1f GB/s\n", rep, dt * 1e3, total / dt / 1e9);
  }
  {
    void* pin = nullptr;
    double t0 = now();
    OK(hipHostMalloc(&pin, total, hipHostMallocDefault));
    double t_alloc = now() - t0;
    t0 = now();
    memcpy(pin, src, total);
    double t_cp = now() - t0;
    for (int rep = 0; rep < 2; rep++) {
      t0 = now();
      OK(hipMemcpyAsync(dst, pin, total, hipMemcpyHostToDevice, st));
      OK(hipStreamSynchronize(st));
      double dt = now() - t0;
      printf("B  hipMemcpyAsync from pinned (rep %d):         %7.1f ms  %6.1f GB/s\n", rep, dt * 1e3, total / dt / 1e9);
    }
    printf("   (hipHostMalloc of %zu MiB: %.1f ms; single-thread memcpy into it: %.1f ms = %.1f GB/s)\n", total >> 20, t_alloc * 1e3, t_cp * 1e3,
           total / t_cp / 1e9);
    OK(hipHostFree(pin));
  }
  // C: ring of pinned slices, T copy threads
  const int RING = 8;
  std::vector<void*> ring(RING);
  {
    double t0 = now();
    for (auto& p : ring) OK(hipHostMalloc(&p, slice, hipHostMallocDefault));
    printf("E  hipHostMalloc of the %d x %zu MiB ring: %.1f ms\n", RING, slice >> 20, (now() - t0) * 1e3);
  }
  std::vector<hipEvent_t> done(RING);
  for (auto& e : done) OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const size_t nslices = (total + slice - 1) / slice;
  for (int T : {1, 2, 4, 8, 16}) {
    for (int rep = 0; rep < 2; rep++) {
      double t0 = now();
      for (size_t s = 0; s < nslices; s++) {
        const int r = (int)(s % RING);
        if (s >= (size_t)RING) OK(hipEventSynchronize(done[r]));
        const size_t off = s * slice, len = std::min(slice, total - off);
        if (T == 1) {
          memcpy(ring[r], src + off, len);
        } else {
          std::vector<std::thread> th;
          const size_t per = (len / T + 4095) & ~(size_t)4095;
          for (int t = 0; t < T; t++) {
            const size_t a = std::min(len, per * t), b = std::min(len, a + per);
            if (b > a) th.emplace_back([=, &ring] { memcpy((uint8_t*)ring[r] + a, src + off + a, b - a); });
          }
          for (auto& x : th) x.join();
        }
        OK(hipMemcpyAsync((uint8_t*)dst + off, ring[r], len, hipMemcpyHostToDevice, st));
        OK(hipEventRecord(done[r], st));
      }
      OK(hipStreamSynchronize(st));
      double dt = now() - t0;
      if (rep) printf("C  %2d copy thread(s) -> pinned ring -> DMA:       %7.1f ms  %6.1f GB/s\n", T, dt * 1e3, total / dt / 1e9);
    }
  }
  // D: register slices in place
  for (size_t reg_slice : {slice, slice * 4, slice * 16}) {
    if (reg_slice > total) break;
    double t0 = now(), t_reg = 0, t_unreg = 0;
    const size_t ns = (total + reg_slice - 1) / reg_slice;
    for (size_t s = 0; s < ns; s++) {
      const size_t off = s * reg_slice, len = std::min(reg_slice, total - off);
      double a = now();
      OK(hipHostRegister(src + off, len, hipHostRegisterDefault));
      t_reg += now() - a;
      OK(hipMemcpyAsync((uint8_t*)dst + off, src + off, len, hipMemcpyHostToDevice, st));
      OK(hipStreamSynchronize(st));
      a = now();
      OK(hipHostUnregister(src + off));
      t_unreg += now() - a;
    }
    double dt = now() - t0;
    printf("D  hipHostRegister %4zu-MiB slices in place (serial): %7.1f ms  %6.1f GB/s  (register %.1f ms, unregister %.1f ms)\n", reg_slice >> 20,
           dt * 1e3, total / dt / 1e9, t_reg * 1e3, t_unreg * 1e3);
  }
  {
    // D2: register on a helper thread one slice ahead of the DMA
    const size_t reg_slice = slice * 4;
    const size_t ns = (total + reg_slice - 1) / reg_slice;
    std::vector<std::atomic<int>> ready(ns);
    for (auto& r : ready) r = 0;
    double t0 = now();
    std::thread reg([&] {
      for (size_t s = 0; s < ns; s++) {
        const size_t off = s * reg_slice, len = std::min(reg_slice, total - off);
        OK(hipHostRegister(src + off, len, hipHostRegisterDefault));
        ready[s] = 1;
      }
    });
    for (size_t s = 0; s < ns; s++) {
      while (!ready[s]) std::this_thread::yield();
      const size_t off = s * reg_slice, len = std::min(reg_slice, total - off);
      OK(hipMemcpyAsync((uint8_t*)dst + off, src + off, len, hipMemcpyHostToDevice, st));
    }
    OK(hipStreamSynchronize(st));
    double dt = now() - t0;
    reg.join();
    double t1 = now();
    for (size_t s = 0; s < ns; s++) OK(hipHostUnregister(src + s * reg_slice));
    printf("D2 register ahead on a helper thread (%zu-MiB slices): %7.1f ms  %6.1f GB/s  (+ unregister afterwards %.1f ms)\n", reg_slice >> 20, dt * 1e3,
           total / dt / 1e9, (now() - t1) * 1e3);
  }
  // F: hipMalloc / hipFree cost of large buffers (a stateless call allocates ~45 GB and gives it back)
  for (size_t gb : {1, 8, 32}) {
    void* p = nullptr;
    double t0 = now();
    OK(hipMalloc(&p, gb << 30));
    double ta = now() - t0;
    t0 = now();
    OK(hipMemsetAsync(p, 0, gb << 30, st));
    OK(hipStreamSynchronize(st));
    double tm = now() - t0;
    t0 = now();
    OK(hipFree(p));
    printf("F  hipMalloc %2zu GiB: %.2f ms, first memset %.2f ms, hipFree %.2f ms\n", gb, ta * 1e3, tm * 1e3, (now() - t0) * 1e3);
  }
  return 0;
}
