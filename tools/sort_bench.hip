// Radix-sort tuning probe: sorts E (key, value) u32 pairs on `keybits` bits with rocPRIM onesweep at several radix widths.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill(uint32_t* k, uint32_t* v, size_t n, uint32_t mask) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) { uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; k[i] = x & mask; v[i] = (uint32_t)i; }
}

template <class Config>
float run(const char* name, uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, size_t n, unsigned keybits) {
  rocprim::double_buffer<uint32_t> kb(k0, k1), vb(v0, v1);
  size_t tmp = 0;
  CHECK((rocprim::radix_sort_pairs<Config>(nullptr, tmp, kb, vb, n, 0, keybits, 0)));
  void* d = nullptr; CHECK(hipMalloc(&d, tmp ? tmp : 16));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    fill<<<(n + 255) / 256, 256>>>(k0, v0, n, (1u << keybits) - 1);
    rocprim::double_buffer<uint32_t> kb2(k0, k1), vb2(v0, v1);
    CHECK(hipEventRecord(e0));
    CHECK((rocprim::radix_sort_pairs<Config>(d, tmp, kb2, vb2, n, 0, keybits, 0)));
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-28s keybits=%u n=%zu: %.3f ms  (%.1f Gpairs/s, tmp %zu MB)\n", name, keybits, n, best, n / best / 1e6, tmp >> 20);
  CHECK(hipFree(d));
  return best;
}

// The same entries as ONE u64 per entry (key in bits 32.., value below): keys-only sort on the key bits.
__global__ void fill64(uint64_t* k, size_t n, uint32_t mask) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) { uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; k[i] = ((uint64_t)(x & mask) << 32) | (uint32_t)i; }
}
float run64(uint64_t* k0, uint64_t* k1, size_t n, unsigned keybits) {
  rocprim::double_buffer<uint64_t> kb(k0, k1);
  size_t tmp = 0;
  CHECK(rocprim::radix_sort_keys(nullptr, tmp, kb, n, 32, 32 + keybits, 0));
  void* d = nullptr; CHECK(hipMalloc(&d, tmp ? tmp : 16));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    fill64<<<(n + 255) / 256, 256>>>(k0, n, (1u << keybits) - 1);
    rocprim::double_buffer<uint64_t> kb2(k0, k1);
    CHECK(hipEventRecord(e0));
    CHECK(rocprim::radix_sort_keys(d, tmp, kb2, n, 32, 32 + keybits, 0));
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-28s keybits=%u n=%zu: %.3f ms  (%.1f Gentries/s, tmp %zu MB)\n", "u64 keys-only (key<<32|val)", keybits, n, best, n / best / 1e6, tmp >> 20);
  CHECK(hipFree(d));
  return best;
}

template <unsigned B, unsigned IPT = 12>
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<256, IPT>, B>>;

int main(int argc, char** argv) {
  size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : (size_t)13 << 26;
  unsigned keybits = argc > 2 ? atoi(argv[2]) : 24;
  uint32_t *k0, *k1, *v0, *v1;
  CHECK(hipMalloc(&k0, n * 4)); CHECK(hipMalloc(&k1, n * 4)); CHECK(hipMalloc(&v0, n * 4)); CHECK(hipMalloc(&v1, n * 4));
  run<rocprim::default_config>("default", k0, k1, v0, v1, n, keybits);
  run<cfg<6>>("onesweep 6 bits", k0, k1, v0, v1, n, keybits);
  run<cfg<7>>("onesweep 7 bits", k0, k1, v0, v1, n, keybits);
  run<cfg<8>>("onesweep 8 bits", k0, k1, v0, v1, n, keybits);
  run<cfg<8, 16>>("onesweep 8 bits ipt16", k0, k1, v0, v1, n, keybits);
  run<cfg<8, 20>>("onesweep 8 bits ipt20", k0, k1, v0, v1, n, keybits);
  run64(reinterpret_cast<uint64_t*>(k0), reinterpret_cast<uint64_t*>(v0), n / 2 * 2 == n ? n / 2 : n / 2, keybits);   // (half the entries: the two u32 arrays hold n/2 u64)
  // (rocPRIM onesweep needs 2^bits-proportional LDS: 9 bits already asks for 262 KB > the 160 KB of a gfx950 CU)
  return 0;
}
