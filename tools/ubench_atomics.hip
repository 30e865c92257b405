// ubench_atomics.hip -- throughput of device-scope 32-bit atomics on MI355X, the bookkeeping primitive of a bucket partition:
//   * global_atomic_add (no return / returning) to random counters in tables of 2^10 .. 2^23 entries, all CUs issuing;
//   * the same with a block's updates first combined in LDS (ds_add_u32) and flushed once per block;
//   * scattered stores of R-byte runs into random bins (how short may a partition's per-bin run be before HBM write
//     efficiency collapses?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/ubench_atomics.hip -o tools/ubench_atomics && tools/ubench_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <bool RET>
__global__ void __launch_bounds__(256) k_atomic(uint32_t* __restrict__ table, uint32_t mask, uint32_t iters, uint32_t* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  uint32_t s = t * 2654435761u + 1u, acc = 0;
  for (uint32_t k = 0; k < iters; k++) {
    s = mix(s + k);
    if (RET)
      acc += atomicAdd(&table[s & mask], 1u);
    else
      __hip_atomic_fetch_add(&table[s & mask], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (RET) out[t] = acc;
}

// R-byte runs scattered to random bins: lane group of R/16 lanes writes one contiguous run; bins advance by R per visit.
template <int R>
__global__ void __launch_bounds__(256) k_scatter_runs(uint4* __restrict__ dst, uint32_t nbins, uint32_t bin_bytes, uint32_t iters) {
  constexpr int G = R / 16;                       // lanes per run
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t grp = t / G, sub = t % G;
  uint32_t s = grp * 2654435761u + 7u;
  for (uint32_t k = 0; k < iters; k++) {
    s = mix(s + k);
    const uint32_t bin = s % nbins;
    const uint32_t slot = (mix(s ^ 0x9e3779b9u) % (bin_bytes / R));   // a run-aligned slot inside the bin's region
    uint4 v = {s, k, t, bin};
    dst[((size_t)bin * bin_bytes + (size_t)slot * R) / 16 + sub] = v;
  }
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device=%s CUs=%d\n", prop.gcnArchName, prop.multiProcessorCount);
  uint32_t *table, *out;
  const size_t TMAX = (size_t)1 << 24;
  CHECK(hipMalloc(&table, TMAX * 4)); CHECK(hipMalloc(&out, (size_t)1 << 26));
  CHECK(hipMemset(table, 0, TMAX * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const uint32_t blocks = 256 * 8, iters = 512;
  printf("%-44s %10s %12s\n", "random 32-bit atomics, 8 blocks/CU", "ms", "G atomics/s");
  for (int ret = 0; ret < 2; ret++)
    for (int lg : {10, 13, 16, 19, 21, 23}) {
      const uint32_t mask = (1u << lg) - 1;
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        if (ret) hipLaunchKernelGGL(k_atomic<true>, dim3(blocks), dim3(256), 0, 0, table, mask, iters, out);
        else hipLaunchKernelGGL(k_atomic<false>, dim3(blocks), dim3(256), 0, 0, table, mask, iters, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      char name[64]; snprintf(name, sizeof name, "%s table 2^%d counters", ret ? "returning" : "no-return", lg);
      printf("%-44s %10.3f %12.2f\n", name, best, (double)blocks * 256 * iters / best / 1e6);
    }
  // scattered runs
  uint4* dst; const size_t DST = (size_t)8 << 30;
  CHECK(hipMalloc(&dst, DST));
  printf("%-44s %10s %12s\n", "scattered R-byte runs (8 GB target)", "ms", "GB/s");
  auto run_sc = [&](auto kern, int R, uint32_t nbins) {
    const uint32_t bin_bytes = (uint32_t)(DST / nbins);
    float best = 1e30f;
    const uint32_t it = 256;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dst, nbins, bin_bytes, it);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    char name[64]; snprintf(name, sizeof name, "run %4d B, %u bins", R, nbins);
    printf("%-44s %10.3f %12.1f\n", name, best, (double)blocks * 256 * it * 16 / best / 1e6);
  };
  for (uint32_t nb : {8192u, 65536u}) {
    run_sc(k_scatter_runs<16>, 16, nb);
    run_sc(k_scatter_runs<32>, 32, nb);
    run_sc(k_scatter_runs<64>, 64, nb);
    run_sc(k_scatter_runs<128>, 128, nb);
    run_sc(k_scatter_runs<256>, 256, nb);
  }
  return 0;
}
