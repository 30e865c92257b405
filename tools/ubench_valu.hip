// Micro-benchmark of the gfx950 VALU instructions the 384-bit Montgomery kernels are built from.
// Prints, per instruction, the sustained issue cost in cycles per wave64-instruction per SIMD
// (assuming the 2.4 GHz max clock; v_add_u32 is the full-rate calibration row).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

#define REP4(s) s s s s
#define REP16(s) REP4(REP4(s))
#define REP64(s) REP4(REP16(s))

// 4 independent chains, 64 instr per chain-group per loop iteration => 256 instrs / iter.
#define KERNEL(name, body)                                                                 \
  __global__ void __launch_bounds__(256) name(uint32_t* out, int iters) {                  \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                    \
    uint32_t b = threadIdx.x * 2654435761u + 12345u, c = blockIdx.x | 1u;                 \
    uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3, wb = ((uint64_t)b << 32) | c;             \
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = 1.0000001, dc = 0.5;                   \
    for (int i = 0; i < iters; i++) { REP64(body) }                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(w0 + w1 + w2 + w3) + (uint32_t)(d0 + d1 + d2 + d3); \
  }

KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
KERNEL(k_mov_b32, asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
KERNEL(k_add_co, asm volatile("v_add_co_u32 %0, vcc, %0, %4\n\tv_add_co_u32 %1, vcc, %1, %4\n\tv_add_co_u32 %2, vcc, %2, %4\n\tv_add_co_u32 %3, vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
KERNEL(k_addc_co, asm volatile("v_addc_co_u32 %0, vcc, %0, %4, vcc\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, %2, %4, vcc\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
KERNEL(k_mul_lo_u32, asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
KERNEL(k_mul_hi_u32, asm volatile("v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
KERNEL(k_mad_u32_u24, asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n\tv_mad_u32_u24 %1, %1, %4, %5\n\tv_mad_u32_u24 %2, %2, %4, %5\n\tv_mad_u32_u24 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
KERNEL(k_mul_hi_u32_u24, asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n\tv_mul_hi_u32_u24 %1, %1, %4\n\tv_mul_hi_u32_u24 %2, %2, %4\n\tv_mul_hi_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
KERNEL(k_mad_u64_u32, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_mad_u64_u32_addc, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(w0), "+v"(w1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_mad_u64_u32_dep, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\tv_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(wb));)
KERNEL(k_fma_f64, asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));)
KERNEL(k_mul_f64, asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));)
KERNEL(k_add_f64, asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dc));)
KERNEL(k_fma_f32, asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
KERNEL(k_mad_i32_i24, asm volatile("v_mad_i32_i24 %0, %0, %4, %5\n\tv_mad_i32_i24 %1, %1, %4, %5\n\tv_mad_i32_i24 %2, %2, %4, %5\n\tv_mad_i32_i24 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %4, 7\n\tv_alignbit_b32 %1, %1, %4, 7\n\tv_alignbit_b32 %2, %2, %4, 7\n\tv_alignbit_b32 %3, %3, %4, 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
KERNEL(k_lshrrev_b64, asm volatile("v_lshrrev_b64 %0, 28, %0\n\tv_lshrrev_b64 %1, 28, %1\n\tv_lshrrev_b64 %2, 28, %2\n\tv_lshrrev_b64 %3, 28, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));)
KERNEL(k_and_b32, asm volatile("v_and_b32 %0, %0, %4\n\tv_and_b32 %1, %1, %4\n\tv_and_b32 %2, %2, %4\n\tv_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
KERNEL(k_sub_u32, asm volatile("v_sub_u32 %0, %0, %4\n\tv_sub_u32 %1, %1, %4\n\tv_sub_u32 %2, %2, %4\n\tv_sub_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
KERNEL(k_lshrrev_b32, asm volatile("v_lshrrev_b32 %0, 3, %0\n\tv_lshrrev_b32 %1, 3, %1\n\tv_lshrrev_b32 %2, 3, %2\n\tv_lshrrev_b32 %3, 3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_add3_u32, asm volatile("v_add3_u32 %0, %0, %4, %5\n\tv_add3_u32 %1, %1, %4, %5\n\tv_add3_u32 %2, %2, %4, %5\n\tv_add3_u32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
KERNEL(k_and_or_b32, asm volatile("v_and_or_b32 %0, %0, %4, %5\n\tv_and_or_b32 %1, %1, %4, %5\n\tv_and_or_b32 %2, %2, %4, %5\n\tv_and_or_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
KERNEL(k_cndmask_sgpr, asm volatile("v_cndmask_b32 %0, %0, %4, s[10:11]\n\tv_cndmask_b32 %1, %1, %4, s[10:11]\n\tv_cndmask_b32 %2, %2, %4, s[10:11]\n\tv_cndmask_b32 %3, %3, %4, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s10", "s11");)
KERNEL(k_mad_u64_u32_sconst, asm volatile("v_mad_u64_u32 %0, vcc, s20, %5, %0\n\tv_mad_u64_u32 %1, vcc, s20, %5, %1\n\tv_mad_u64_u32 %2, vcc, s20, %5, %2\n\tv_mad_u64_u32 %3, vcc, s20, %5, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b), "v"(c) : "vcc", "s20");)
KERNEL(k_mad_u64_u32_sgpr, asm volatile("v_mad_u64_u32 %0, s[10:11], %4, %5, %0\n\tv_mad_u64_u32 %1, s[12:13], %4, %5, %1\n\tv_mad_u64_u32 %2, s[10:11], %4, %5, %2\n\tv_mad_u64_u32 %3, s[12:13], %4, %5, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b), "v"(c) : "s10", "s11", "s12", "s13");)

KERNEL(k_cndmask_e64_vcc, asm volatile("v_cndmask_b32_e64 %0, %0, %4, vcc\n\tv_cndmask_b32_e64 %1, %1, %4, vcc\n\tv_cndmask_b32_e64 %2, %2, %4, vcc\n\tv_cndmask_b32_e64 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
KERNEL(k_cndmask_cmp, asm volatile("v_cmp_gt_u32_e32 vcc, %4, %0\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %4, vcc\n\tv_cndmask_b32_e32 %2, %2, %4, vcc\n\tv_cndmask_b32_e32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
KERNEL(k_cndmask_distinct, asm volatile("v_cndmask_b32_e32 %0, %5, %4, vcc\n\tv_cndmask_b32_e32 %1, %5, %4, vcc\n\tv_cndmask_b32_e32 %2, %5, %4, vcc\n\tv_cndmask_b32_e32 %3, %5, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_xad_u32, asm volatile("v_xad_u32 %0, %0, %4, %5\n\tv_xad_u32 %1, %1, %4, %5\n\tv_xad_u32 %2, %2, %4, %5\n\tv_xad_u32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
KERNEL(k_bfi_b32, asm volatile("v_bfi_b32 %0, %0, %4, %5\n\tv_bfi_b32 %1, %1, %4, %5\n\tv_bfi_b32 %2, %2, %4, %5\n\tv_bfi_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
KERNEL(k_mad_nop_shift, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\ts_nop 0\n\tv_lshrrev_b64 %1, 28, %0\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\ts_nop 0\n\tv_lshrrev_b64 %3, 28, %2" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_mad_shift, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_lshrrev_b64 %1, 28, %0\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_lshrrev_b64 %3, 28, %2" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b), "v"(c) : "vcc");)

typedef void (*kern_t)(uint32_t*, int);
struct Row { const char* name; kern_t k; };

int main(int argc, char** argv) {
  int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
  int iters = argc > 2 ? atoi(argv[2]) : 2000;
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double ghz = prop.clockRate / 1e6;
  printf("device=%s arch=%s CUs=%d clockRate=%.3f GHz waves_per_simd=%d iters=%d\n", prop.name, prop.gcnArchName, cus, ghz, waves_per_simd, iters);
  int blocks = cus * waves_per_simd;      // 256 threads = 4 waves = 1 wave per SIMD per block
  uint32_t* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  Row rows[] = {
    {"v_add_u32", k_add_u32}, {"v_mov_b32", k_mov_b32}, {"v_add_co_u32", k_add_co}, {"v_addc_co_u32", k_addc_co},
    {"v_cndmask_b32", k_cndmask}, {"v_alignbit_b32", k_alignbit}, {"v_fma_f32", k_fma_f32},
    {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mad_i32_i24", k_mad_i32_i24},
    {"v_mul_hi_u32_u24", k_mul_hi_u32_u24},
    {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mad_u64_u32(sgpr carry)", k_mad_u64_u32_sgpr}, {"v_mad_u64_u32 dependent", k_mad_u64_u32_dep},
    {"v_mad_u64_u32+v_addc (pair=2 instr)", k_mad_u64_u32_addc},
    {"v_lshl_add_u64", k_lshl_add_u64}, {"v_lshrrev_b64", k_lshrrev_b64}, {"v_and_b32", k_and_b32}, {"v_sub_u32", k_sub_u32},
    {"v_lshrrev_b32", k_lshrrev_b32}, {"v_add3_u32", k_add3_u32}, {"v_and_or_b32", k_and_or_b32}, {"v_cndmask_b32 (sgpr mask)", k_cndmask_sgpr},
    {"v_mad_u64_u32 (sgpr operand)", k_mad_u64_u32_sconst},
    {"v_cndmask_b32_e64 (vcc mask)", k_cndmask_e64_vcc}, {"v_cmp + 4 v_cndmask_e32 (5 instr)", k_cndmask_cmp},
    {"v_cndmask_b32_e32 dst != src", k_cndmask_distinct}, {"v_xad_u32", k_xad_u32}, {"v_bfi_b32", k_bfi_b32},
    {"mad,s_nop,lshr64 x2 (6 instr, 4 VALU)", k_mad_nop_shift}, {"mad,lshr64 x2 (4 instr)", k_mad_shift}, {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64},
  };
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-40s %10s %14s %16s\n", "instruction", "ms", "cyc/wave-instr", "Glane-ops/s");
  for (auto& r : rows) {
    hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, out, 10);   // warm
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, out, iters);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double instr_per_wave = (double)iters * 256.0;
    double cyc = best * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd);   // per SIMD: waves_per_simd waves share it
    double lane_ops = instr_per_wave * 64.0 * blocks * 4 / (best * 1e-3) / 1e9;
    printf("%-40s %10.3f %14.2f %16.1f\n", r.name, best, cyc, lane_ops);
  }
  return 0;
}
