// ubench_batched_affine.hip -- row f4 of SURVEY.md section 8, closed by measurement: is batched-affine addition (the
// `batch_affine_addition` of ARK ec/src/models/short_weierstrass.rs:239-319: chord-and-tangent in affine coordinates with
// ONE inversion per batch through Montgomery's trick) faster on this chip than the 7M extended twisted-Edwards mixed
// addition the accumulation uses?
//
// Each lane performs B independent affine additions R_i = P_i + Q_i (P_i != +-Q_i), operands and results in HBM in a
// lane-interleaved layout so that every access of a wave is coalesced (the friendliest possible memory pattern: a real
// bucket accumulation would gather):
//   forward   d_i = x2_i - x1_i,  pre_i = d_0 ... d_i                      (1 multiplication, one 56-B store)
//   invert    inv = pre_{B-1}^(p-2)                                        (~570 multiplications, amortised over B)
//   backward  dinv = inv * pre_{i-1};  inv *= d_i;  lambda = (y2 - y1) dinv;  x3 = lambda^2 - x1 - x2;  y3 = lambda (x1 - x3) - y1
//                                                                          (4 multiplications + 1 squaring)
// i.e. 5M + 1S + 570/B per addition against 7M -- but 5 x 56-B operand reads twice, a prefix store + load and a result
// store per addition against ONE 192-B gather.  Field arithmetic: fp28.hpp, the same code as the engine.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/ubench_batched_affine.hip -o tools/ubench_batched_affine
#include "../2022-entries_amd/csrc/laws.hpp"
#include <cstdio>
#include <cstdlib>
using namespace msm;
using F = Bls12_377_Fq;

// element (i, lane t) of an array of B x T field elements lives at [i * T + t]: coalesced across the wave
__global__ void __launch_bounds__(256, 2) k_batched_affine(const Fe* __restrict__ x1, const Fe* __restrict__ y1, const Fe* __restrict__ x2,
                                                           const Fe* __restrict__ y2, Fe* __restrict__ pre, Fe* __restrict__ x3o,
                                                           Fe* __restrict__ y3o, int B, size_t T) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  Modulus<F> md;
  Fe run;
  fe_set(run, F::ONE);
  for (int i = 0; i < B; i++) {
    const Fe a = x1[i * T + t], b = x2[i * T + t];
    Fe d;
    fe_sub(d, b, a, F::BIAS2_28);
    pre[i * T + t] = run;               // product of the earlier differences
    fe_mul<F>(run, run, d, md);
  }
  Fe inv;
  fe_inv<F>(inv, run, md);
  for (int i = B - 1; i >= 0; i--) {
    const Fe ax = x1[i * T + t], ay = y1[i * T + t], bx = x2[i * T + t], by = y2[i * T + t];
    const Fe pr = pre[i * T + t];
    Fe d, dinv, dy, lam, l2, s, x3, y3, t1;
    fe_sub(d, bx, ax, F::BIAS2_28);
    fe_mul<F>(dinv, inv, pr, md);
    fe_mul<F>(inv, inv, d, md);
    fe_sub(dy, by, ay, F::BIAS2_28);
    fe_mul<F>(lam, dy, dinv, md);
    fe_sqr<F>(l2, lam, md);
    fe_add(s, ax, bx);
    fe_sub(x3, l2, s, F::BIAS4_29);
    fe_carry(x3);
    fe_sub(t1, ax, x3, F::BIAS8_29);
    fe_carry(t1);
    fe_mul<F>(y3, lam, t1, md);
    fe_sub(y3, y3, ay, F::BIAS2_28);
    fe_carry(y3);
    x3o[i * T + t] = x3;
    y3o[i * T + t] = y3;
  }
}

// the comparison arm at the same occupancy: B twisted-Edwards mixed additions per lane, bases streamed coalesced from HBM
__global__ void __launch_bounds__(256, 2) k_te_stream(const Fe* __restrict__ x1, const Fe* __restrict__ y1, const Fe* __restrict__ x2,
                                                      Xyzz* __restrict__ out, int B, size_t T) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  Modulus<F> md;
  Xyzz acc;
  te_set_identity<F>(acc);
  for (int i = 0; i < B; i++) {
    TeAffine p;
    p.ymx = x1[i * T + t];
    p.ypx = y1[i * T + t];
    p.td = x2[i * T + t];
    te_madd<F, true>(acc, p, (i & 1) != 0, md);
  }
  out[t] = acc;
}

__global__ void k_fill(Fe* dst, size_t n, uint32_t seed) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s = (uint32_t)i * 2654435761u + seed;
  Fe v;
  for (int j = 0; j < NL; j++) {
    s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s += 0x9e3779b9u;
    v.v[j] = (s & LMASK) >> (j == NL - 1 ? 16 : 0);   // values < p: the top limb is kept small
  }
  dst[i] = v;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
  const int blocks = 256 * 2 * 2;
  const size_t T = (size_t)blocks * 256;
  const int BMAX = 1024;
  Fe *x1, *y1, *x2, *y2, *pre, *x3, *y3; Xyzz* acc;
  const size_t bytes = (size_t)BMAX * T * sizeof(Fe);
  for (Fe** p : {&x1, &y1, &x2, &y2, &pre, &x3, &y3}) CHECK(hipMalloc(p, bytes));
  CHECK(hipMalloc(&acc, T * sizeof(Xyzz)));
  // canonical-looking pseudo-random limbs; only the instruction and memory streams matter
  uint32_t seed = 1;
  for (Fe* dst : {x1, y1, x2, y2}) {
    const size_t n = (size_t)BMAX * T;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dst, n, seed++ * 7919u);
  }
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%zu lanes (2 waves/SIMD), BLS12-377 base field, operands streamed from HBM (coalesced)\n", T);
  printf("%-44s %10s %14s %12s\n", "kernel", "ms", "G additions/s", "GB/s moved");
  for (int B : {64, 256, 1024}) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_batched_affine, dim3(blocks), dim3(256), 0, 0, x1, y1, x2, y2, pre, x3, y3, B, T);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    char name[64]; snprintf(name, sizeof name, "batched affine, B = %d per lane", B);
    // bytes per addition: forward x1,x2 (112) + pre store (56); backward 4 operands (224) + pre load (56) + result (112)
    printf("%-44s %10.3f %14.3f %12.1f\n", name, best, (double)T * B / best / 1e6, (double)T * B * 560 / best / 1e6);
  }
  for (int B : {64, 1024}) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_te_stream, dim3(blocks), dim3(256), 0, 0, x1, y1, x2, acc, B, T);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    char name[64]; snprintf(name, sizeof name, "twisted Edwards madd (7M), %d per lane", B);
    printf("%-44s %10.3f %14.3f %12.1f\n", name, best, (double)T * B / best / 1e6, (double)T * B * 168 / best / 1e6);
  }
  return 0;
}
