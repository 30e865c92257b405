// calib_fetch.hip -- what do rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the ACCESS SHAPES of the accumulate kernel?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/calib_fetch.hip -o tools/calib_fetch
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- tools/calib_fetch      (and once more with --pmc WRITE_SIZE)
//   python tools/calib_fetch_summary.py out                                      -> profiles/r03_calib_fetch.txt
//
// MI355X_MICROARCH.md: FETCH_SIZE = TCC_EA0_RDREQ x 64 B, so a wide coalesced stream (128-B requests) reads 0.5x; every other
// shape is "uncalibrated: calibrate on a known byte count in your own access pattern".  Each kernel below moves an exactly
// known number of bytes, every byte once (tables far larger than the 256-MB Infinity Cache, every record visited at most once):
//   calib_stream16        16 B per lane, coalesced                         (the guide's reference case)
//   calib_stream8         8 B per lane, coalesced                          (the sorted-entry stream of k_accumulate_glds)
//   calib_gather_glds<S>  quad-cooperative LDS-DMA gather of S x 64-B records, S = 2, 3, 4   (XYZZ bases 128 B, twisted-Edwards
//                         192 B, G2 256 B: global_load_lds_dwordx4, lane l of a quad takes piece l of every sector)
//   calib_gather_lane<S>  one lane per record, S x 64 B by plain 16-B loads: S = 1 is the entry queue's refill (one sector per
//                         lane, lanes 2 KB apart), S = 2..4 the exceptional-pair re-read and k_segreduce
//   calib_write16         16 B per lane, coalesced stores
//   calib_write224        one 224-B bucket record per lane at a random index (seg_flush)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdint.h>

#define OK(e)                                                                     \
  do {                                                                            \
    hipError_t e_ = (e);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__global__ void __launch_bounds__(256) calib_stream16(const uint4* __restrict__ src, size_t n16, uint4* __restrict__ sink) {
  uint4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 v = src[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;   // never true in practice: keeps the loads alive
}

__global__ void __launch_bounds__(256) calib_stream8(const uint2* __restrict__ src, size_t n8, uint4* __restrict__ sink) {
  uint2 acc = {0, 0};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const uint2 v = src[i];
    acc.x ^= v.x; acc.y += v.y;
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = make_uint4(acc.x, acc.y, 0, 0);
}

// record index visited by (lane t, step k): a bijection on [0, nrec) for nrec a power of two (odd multiplier + offset)
__device__ __forceinline__ uint32_t rec_index(uint32_t t, uint32_t k, uint32_t iters, uint32_t nrec) {
  return ((t * iters + k) * 2654435761u + 40503u) & (nrec - 1);
}

template <int SECT>
__global__ void __launch_bounds__(256) calib_gather_glds(const unsigned char* __restrict__ table, uint32_t nrec, uint32_t iters, uint4* __restrict__ sink) {
  constexpr int RS = 1024, WAVE_LDS = 4 * SECT * RS + 256;
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * WAVE_LDS];
  const uint32_t t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, sub = lane & 3, quad = lane >> 2;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* wave_lds = lds + wave * WAVE_LDS;
  const unsigned char* rec = wave_lds + sub * (SECT * RS + 64) + quad * 64;
  uint4 acc = {0, 0, 0, 0};
  for (uint32_t k = 0; k < iters; k++) {
    const int mine = (int)rec_index(t, k, iters, nrec);
#define ONE(i, ctrl)                                                                                                            \
  do {                                                                                                                          \
    const uint4* s_ = reinterpret_cast<const uint4*>(table + (size_t)(uint32_t)__builtin_amdgcn_update_dpp(0, mine, ctrl, 0xf, 0xf, true) * (SECT * 64)) + sub; \
    _Pragma("unroll") for (int c_ = 0; c_ < SECT; c_++)                                                                         \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(s_ + 4 * c_), (lds_ptr_t)(wave_lds + (i) * (SECT * RS + 64) + c_ * RS), 16, 0, 0); \
  } while (0)
    ONE(0, 0x00);
    ONE(1, 0x55);
    ONE(2, 0xaa);
    ONE(3, 0xff);
#undef ONE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const uint4 v = *reinterpret_cast<const uint4*>(rec + (k % SECT) * RS);
    acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

template <int SECT>
__global__ void __launch_bounds__(256) calib_gather_lane(const unsigned char* __restrict__ table, uint32_t nrec, uint32_t iters, uint4* __restrict__ sink) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  uint4 acc = {0, 0, 0, 0};
  for (uint32_t k = 0; k < iters; k++) {
    const uint4* r = reinterpret_cast<const uint4*>(table + (size_t)rec_index(t, k, iters, nrec) * (SECT * 64));
#pragma unroll
    for (int j = 0; j < SECT * 4; j++) {
      const uint4 v = r[j];
      acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

__global__ void __launch_bounds__(256) calib_write16(uint4* __restrict__ dst, size_t n16) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

__global__ void __launch_bounds__(256) calib_write224(unsigned char* __restrict__ dst, uint32_t nrec, uint32_t iters) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  for (uint32_t k = 0; k < iters; k++) {
    uint4* r = reinterpret_cast<uint4*>(dst + (size_t)rec_index(t, k, iters, nrec) * 224);
#pragma unroll
    for (int j = 0; j < 14; j++) r[j] = make_uint4(t, k, j, 7);
  }
}

int main() {
  const size_t table_bytes = (size_t)12 << 30;   // 12 GiB: 48x the Infinity Cache
  unsigned char* table = nullptr;
  uint4* sink = nullptr;
  OK(hipMalloc(&table, table_bytes));
  OK(hipMalloc(&sink, 4096));
  OK(hipMemset(table, 0x5a, table_bytes));
  OK(hipDeviceSynchronize());
  const uint32_t lanes = 256 * 4 * 64 * 2;   // 2 waves per SIMD, like the accumulate kernel
  const uint32_t blocks = lanes / 256;
  printf("# kernel expected_bytes  (every byte touched once; table %zu MiB)\n", table_bytes >> 20);
  {
    const size_t bytes = (size_t)8 << 30;
    hipLaunchKernelGGL(calib_stream16, dim3(4096), dim3(256), 0, 0, (const uint4*)table, bytes / 16, sink);
    printf("calib_stream16 %zu\n", bytes);
    hipLaunchKernelGGL(calib_stream8, dim3(4096), dim3(256), 0, 0, (const uint2*)table, bytes / 8, sink);
    printf("calib_stream8 %zu\n", bytes);
  }
  {
    // records visited = lanes x iters <= nrec (a bijection: no record twice)
    const uint32_t nrec2 = 1u << 26, nrec3 = 1u << 25, nrec4 = 1u << 25;   // 8 GiB of 128-B, 6 GiB of 192-B, 8 GiB of 256-B records
    const uint32_t it2 = nrec2 / lanes / 2, it3 = nrec3 / lanes, it4 = nrec4 / lanes;
    hipLaunchKernelGGL(calib_gather_glds<2>, dim3(blocks), dim3(256), 0, 0, table, nrec2, it2, sink);
    printf("calib_gather_glds<2> %zu\n", (size_t)lanes * it2 * 128);
    hipLaunchKernelGGL(calib_gather_glds<3>, dim3(blocks), dim3(256), 0, 0, table, nrec3, it3, sink);
    printf("calib_gather_glds<3> %zu\n", (size_t)lanes * it3 * 192);
    hipLaunchKernelGGL(calib_gather_glds<4>, dim3(blocks), dim3(256), 0, 0, table, nrec4, it4, sink);
    printf("calib_gather_glds<4> %zu\n", (size_t)lanes * it4 * 256);
    {
      // the entry queue of k_accumulate_glds: one 64-B sector per lane, four 16-B loads back to back, lanes 2 KB apart
      const uint32_t nrec1 = 1u << 27;   // 8 GiB of 64-B records
      const uint32_t it1 = nrec1 / lanes / 4;
      hipLaunchKernelGGL(calib_gather_lane<1>, dim3(blocks), dim3(256), 0, 0, table, nrec1, it1, sink);
      printf("calib_gather_lane<1> %zu\n", (size_t)lanes * it1 * 64);
    }
    hipLaunchKernelGGL(calib_gather_lane<2>, dim3(blocks), dim3(256), 0, 0, table, nrec2, it2 / 4, sink);
    printf("calib_gather_lane<2> %zu\n", (size_t)lanes * (it2 / 4) * 128);
    hipLaunchKernelGGL(calib_gather_lane<3>, dim3(blocks), dim3(256), 0, 0, table, nrec3, it3 / 4, sink);
    printf("calib_gather_lane<3> %zu\n", (size_t)lanes * (it3 / 4) * 192);
    hipLaunchKernelGGL(calib_gather_lane<4>, dim3(blocks), dim3(256), 0, 0, table, nrec4, it4 / 4, sink);
    printf("calib_gather_lane<4> %zu\n", (size_t)lanes * (it4 / 4) * 256);
  }
  {
    const size_t bytes = (size_t)8 << 30;
    hipLaunchKernelGGL(calib_write16, dim3(4096), dim3(256), 0, 0, (uint4*)table, bytes / 16);
    printf("calib_write16 %zu\n", bytes);
    const uint32_t nrec = 1u << 25;   // 7 GiB of 224-B records
    const uint32_t it = nrec / lanes / 4;
    hipLaunchKernelGGL(calib_write224, dim3(blocks), dim3(256), 0, 0, table, nrec, it);
    printf("calib_write224 %zu\n", (size_t)lanes * it * 224);
  }
  OK(hipDeviceSynchronize());
  OK(hipGetLastError());
  return 0;
}
