// ubench_gather.hip -- how many random fixed-size records per second can the chip gather?  One record per lane per
// iteration (the access pattern of k_accumulate: every lane reads a different, random base), record sizes 64..256 B,
// table 2^26 records, almost no compute.  This is the memory-side ceiling the bucket accumulation lives under.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/ubench_gather.hip -o tools/ubench_gather && tools/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

template <int BYTES, int STRIDE>
__global__ void __launch_bounds__(256) kgather(const uint4* __restrict__ table, uint32_t nrec, uint32_t iters, uint4* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  uint32_t s = t * 2654435761u + 12345u;
  uint4 acc = {0, 0, 0, 0};
  for (uint32_t k = 0; k < iters; k++) {
    s = s * 1664525u + 1013904223u;
    const uint32_t idx = (s >> 4) % nrec;
    const uint4* r = table + (size_t)idx * (STRIDE / 16);
#pragma unroll
    for (int j = 0; j < BYTES / 16; j++) {
      const uint4 v = r[j];
      acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
  }
  out[t] = acc;
}

// Quad-cooperative variant: the four lanes of a quad fetch their four records together, lane l taking bytes [16 l, 16 l + 16)
// of every 64-B sector, so one wave-instruction touches 16 cache lines instead of 64.
template <int BYTES, int STRIDE>
__global__ void __launch_bounds__(256) kgather_quad(const uint4* __restrict__ table, uint32_t nrec, uint32_t iters, uint4* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t sub = threadIdx.x & 3;
  uint32_t s = t * 2654435761u + 12345u;
  uint4 acc = {0, 0, 0, 0};
  for (uint32_t k = 0; k < iters; k++) {
    s = s * 1664525u + 1013904223u;
    const uint32_t idx = (s >> 4) % nrec;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t idx_i = __shfl(idx, (threadIdx.x & 60) + i, 64);
      const uint4* r = table + (size_t)idx_i * (STRIDE / 16) + sub;
#pragma unroll
      for (int c = 0; c < (BYTES + 63) / 64; c++) {
        const uint4 v = r[4 * c];
        acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
      }
    }
  }
  out[t] = acc;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1); } } while (0)

template <int BYTES, int STRIDE, bool QUAD = false>
void run(const uint4* table, uint32_t nrec, uint4* out, int waves_per_simd) {
  const uint32_t lanes = 256 * 4 * 64 * waves_per_simd, iters = 2048;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    CHECK(hipEventRecord(e0));
    if (QUAD) kgather_quad<BYTES, STRIDE><<<lanes / 256, 256>>>(table, nrec, iters, out);
    else kgather<BYTES, STRIDE><<<lanes / 256, 256>>>(table, nrec, iters, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double recs = (double)lanes * iters;
  printf("%s record %3d B (stride %3d B), %d waves/SIMD: %7.2f G records/s   %6.2f TB/s of payload   %6.2f TB/s of 64-B sectors touched\n", QUAD ? "quad-cooperative" : "one lane each   ", BYTES, STRIDE,
         waves_per_simd, recs / ms / 1e6, recs * BYTES / ms / 1e9, recs * ((BYTES + 63) / 64) * 64 / ms / 1e9);
}

int main() {
  const uint32_t nrec = 1u << 26;
  uint4 *table, *out;
  CHECK(hipMalloc(&table, (size_t)nrec * 256));
  CHECK(hipMemset(table, 1, (size_t)nrec * 256));
  CHECK(hipMalloc(&out, (size_t)256 * 4 * 64 * 8 * 16));
  for (int w : {2, 8}) {
    run<64, 64>(table, nrec, out, w);
    run<112, 112>(table, nrec, out, w);
    run<112, 128>(table, nrec, out, w);
    run<128, 128>(table, nrec, out, w);
    run<176, 192>(table, nrec, out, w);
    run<192, 192>(table, nrec, out, w);
    run<176, 256>(table, nrec, out, w);
    run<256, 256>(table, nrec, out, w);
    run<128, 128, true>(table, nrec, out, w);
    run<192, 192, true>(table, nrec, out, w);
    run<256, 256, true>(table, nrec, out, w);
  }
  return 0;
}
