// ubench_madd.hip -- mixed-addition throughput in isolation (operands in registers, no memory traffic) for the group laws:
// XYZZ (8M + 2S, curve.hpp) and extended twisted Edwards (7M, te.hpp) on 14 x 28 and on 13 x 29 limbs (fp28.hpp), at the occupancy
// k_accumulate runs at.  The 28-vs-29 pair is the A/B of profiles/r06_ab_limbs29.txt.
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
    C[i] = acc;
  } else if (V == 1) {
    TeAffine p;
    p.ymx = A[2 * i];
    p.ypx = A[2 * i + 1];
    p.td = A[2 * i];
    te_set_identity<F>(acc);
    for (int k = 0; k < iters; k++) TeLaw<F>::madd(acc, p, (k & 1) != 0, false, md);
    C[i] = acc;
  } else {
    using F29 = Bls12_377_Fq29;
    Modulus<F29> md29;
    TeAffine p;   // (the host fills 28-bit limbs: below every bound of the 29-bit shape too; word 13 is cleared)
    p.ymx = A[2 * i];
    p.ypx = A[2 * i + 1];
    p.ymx.v[13] = p.ypx.v[13] = 0;
    p.td = p.ymx;
    te_set_identity<F29>(acc);
    for (int k = 0; k < iters; k++) TeLaw<F29>::madd(acc, p, (k & 1) != 0, false, md29);
    C[i] = acc;
  }
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
  const char* names[3] = {"XYZZ madd (8M+2S), 14 x 28", "twisted Edwards madd (7M), 14 x 28", "twisted Edwards madd (7M), 13 x 29"};
  for (int round = 0; round < 2; round++)   // twice: the second round runs on a warm (power-throttled) chip, like the product
    for (int v = 0; v < 3; v++) {
      float ms = 0, best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        if (v == 0) kmadd<0><<<blocks, 256>>>(A, C, iters); else if (v == 1) kmadd<1><<<blocks, 256>>>(A, C, iters); else kmadd<2><<<blocks, 256>>>(A, C, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("%-38s %9.3f ms (best of 3: %9.3f)   %.3f G mixed-add/s   %.1f ns per add per lane-slot\n", names[v], ms, best,
             (double)n * iters / ms / 1e6, ms * 1e6 / iters);
    }
  return 0;
}
