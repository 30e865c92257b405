// ubench_madd.hip -- mixed-addition throughput in isolation (operands in registers, no memory traffic) for the two group laws:
// XYZZ (8M + 2S, curve.hpp) and extended twisted Edwards (7M, te.hpp), at the occupancy k_accumulate runs at.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/ubench_madd.hip -o tools/ubench_madd && tools/ubench_madd
#include "../2022-entries_amd/csrc/laws.hpp"
#include <cstdio>
#include <cstdlib>
using namespace msm;
using F = Bls12_377_Fq;

template <int V>
__global__ void __launch_bounds__(256, 2) kmadd(const Fe* A, Xyzz* C, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  Modulus<F> md;
  Xyzz acc;
  if (V == 0) {
    Affine p;
    p.x = A[2 * i];
    p.y = A[2 * i + 1];
    xyzz_from_affine<FpEl<F>>(acc, p, false);
    acc.zz = A[2 * i];   // arbitrary non-trivial ZZ/ZZZ: only the instruction stream matters here
    acc.zzz = A[2 * i + 1];
    for (int k = 0; k < iters; k++) SwLaw<FpEl<F>>::madd(acc, p, (k & 1) != 0, false, md);
  } else {
    TeAffine p;
    p.ymx = A[2 * i];
    p.ypx = A[2 * i + 1];
    p.td = A[2 * i];
    te_set_identity<F>(acc);
    for (int k = 0; k < iters; k++) TeLaw<F>::madd(acc, p, (k & 1) != 0, false, md);
  }
  C[i] = acc;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  const int blocks = 256 * 2 * 4, n = blocks * 256;
  Fe* A; Xyzz* C;
  CHECK(hipMalloc(&A, 2 * n * sizeof(Fe)));
  CHECK(hipMalloc(&C, n * sizeof(Xyzz)));
  Fe* h = (Fe*)malloc(2 * n * sizeof(Fe));
  for (int i = 0; i < 2 * n; i++) for (int j = 0; j < NL; j++) h[i].v[j] = (rand() & LMASK) >> (j == NL - 1 ? 12 : 0);
  CHECK(hipMemcpy(A, h, 2 * n * sizeof(Fe), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int v = 0; v < 2; v++) {
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      if (v == 0) kmadd<0><<<blocks, 256>>>(A, C, iters); else kmadd<1><<<blocks, 256>>>(A, C, iters);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%-28s %9.3f ms   %.3f G mixed-add/s   %.1f ns per add per lane-slot\n", v == 0 ? "XYZZ madd (8M+2S)" : "twisted Edwards madd (7M)", ms,
           (double)n * iters / ms / 1e6, ms * 1e6 / iters);
  }
  return 0;
}
