// ubench_mul.hip -- field-multiplication throughput on gfx950: the header's fe_mul (uninterrupted inline-asm
// v_mad_u64_u32 chains) against the same product scanning written as plain C++ loops (what hipcc schedules by itself).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/ubench_mul.hip -o tools/ubench_mul && tools/ubench_mul
// Results of record: profiles/r01_ubench_mul.txt.
#include "../2022-entries_amd/csrc/fp28.hpp"
#include <cstdio>
#include <cstdlib>
using namespace msm;

template <class F>
__device__ __forceinline__ void fe_mul_plain(Fe& r, const Fe& a, const Fe& b, const Modulus<F>& md) {
  uint32_t m[NL];
  Fe t;
  uint64_t col = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) col += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) col += (uint64_t)m[i] * md.p[k - i];
    m[k] = ((uint32_t)col * F::M0) & LMASK;
    col += (uint64_t)m[k] * md.p[0];
    col >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) col += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) col += (uint64_t)m[i] * md.p[k - i];
    t.v[k - NL] = (uint32_t)col & LMASK;
    col >>= LB;
  }
  t.v[NL - 1] = (uint32_t)col;
  r = t;
}

template <class F, int V>
__global__ void __launch_bounds__(256) kmul(const Fe* A, const Fe* B, Fe* C, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  Modulus<F> md;
  Fe a = A[i], b = B[i];
  for (int k = 0; k < iters; k++) {
    if (V == 0) fe_mul_plain<F>(a, a, b, md);
    else fe_mul<F>(a, a, b, md);
  }
  C[i] = a;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1); } } while (0)

template <class F>
void run(const char* name, int iters) {
  const int blocks = 256 * 8, n = blocks * 256;
  Fe *A, *B, *C0, *C1;
  CHECK(hipMalloc(&A, n * sizeof(Fe))); CHECK(hipMalloc(&B, n * sizeof(Fe)));
  CHECK(hipMalloc(&C0, n * sizeof(Fe))); CHECK(hipMalloc(&C1, n * sizeof(Fe)));
  Fe* h = (Fe*)malloc(n * sizeof(Fe));
  for (int pass = 0; pass < 2; pass++) {
    for (int i = 0; i < n; i++) for (int j = 0; j < NL; j++) h[i].v[j] = (rand() & LMASK) >> (j == NL - 1 ? 12 : 0);
    CHECK(hipMemcpy(pass ? B : A, h, n * sizeof(Fe), hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[2];
  for (int v = 0; v < 2; v++) for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    if (v == 0) kmul<F, 0><<<blocks, 256>>>(A, B, C0, iters); else kmul<F, 1><<<blocks, 256>>>(A, B, C1, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[v], e0, e1);
  }
  Fe* h0 = (Fe*)malloc(n * sizeof(Fe)); Fe* h1 = (Fe*)malloc(n * sizeof(Fe));
  CHECK(hipMemcpy(h0, C0, n * sizeof(Fe), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h1, C1, n * sizeof(Fe), hipMemcpyDeviceToHost));
  int diff = 0;
  for (int i = 0; i < n; i++) for (int j = 0; j < NL; j++) diff += h0[i].v[j] != h1[i].v[j];
  const double muls = (double)n * iters;
  printf("%s iters %6d: plain C++ %8.3f ms (%.2f Gmul/s)   asm chains %8.3f ms (%.2f Gmul/s)   ratio %.3f   mismatching limbs %d\n",
         name, iters, ms[0], muls / ms[0] / 1e6, ms[1], muls / ms[1] / 1e6, ms[1] / ms[0], diff);
}

int main(int argc, char** argv) {
  if (argc > 1) {  // one long run (e.g. 300000 iterations = 2 s per arm) for tools/power_probe.sh
    run<Bls12_377_Fq>("bls12_377 fq", atoi(argv[1]));
    return 0;
  }
  for (int it : {2000, 20000, 60000}) run<Bls12_377_Fq>("bls12_377 fq", it);
  run<Bls12_381_Fq>("bls12_381 fq", 20000);
  return 0;
}
