// msm_kernels.cuh -- the gfx950 kernels of the Pippenger pipeline.
//
//   k_convert_bases   arkworks Affine image (stride bytes, R = 2^384)  ->  device Affine (112 B, radix 2^28, R = 2^392)
//   k_digits          256-bit scalars -> signed c-bit digits -> (bucket key, base index | sign) entries
//   (rocPRIM radix sort of the entries by key -- scaffolding, see msm_engine.hip)
//   k_accumulate      sorted-range walk: each lane owns K consecutive sorted entries and mixed-adds their
//                     bases; finished buckets are stored once, run fragments that cross a lane boundary
//                     go to per-lane head/tail slots
//   k_segreduce       merges the slot fragments (same walk, full XYZZ add), recursively
//   k_bucket_reduce   sum_b b * bucket[b] per window by chunked running sums, recursively
//
// Reference behaviour covered: digit extraction SPK msm/pippenger.cuh:116-123, signed digits
// CMB ProcessSignedDigits.cu:118-151 / P1A mikevoronov sppark/msm/pippenger.cuh:453-479; bucket
// accumulation CMB ComputeBucketSums.cu:139-218, ML msm_kernels.cu:114-142, the sorted-range idea of
// P1A 6block cuda/mypippenger.cu:167-241; bucket reduction SPK msm/pippenger.cuh:210-244,
// CMB ReduceBuckets.cu:77-149.  The decomposition, data layout and balancing are this repo's own
// (DESIGN.md): the walk is balanced per ENTRY, not per bucket, so skewed scalar distributions
// (one hot bucket, the sparse top window) cost the same as uniform ones.
#pragma once
#include "curve.cuh"

namespace msm {

constexpr uint32_t KEY_NONE = 0xffffffffu;
constexpr uint32_t IDX_MASK = 0x7fffffffu;

struct alignas(16) AffineDev {
  Affine p;
};
struct alignas(16) XyzzDev {
  Xyzz p;
};
static_assert(sizeof(AffineDev) == 112, "device affine layout");
static_assert(sizeof(XyzzDev) == 224, "device xyzz layout");

// ------------------------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) k_convert_bases(const uint8_t* __restrict__ in, size_t stride, uint32_t n,
                                                       AffineDev* __restrict__ out, uint8_t* __restrict__ inf) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Modulus<F> md;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(in + (size_t)i * stride);
  uint32_t w[24];
#pragma unroll
  for (int k = 0; k < 24; k++) w[k] = src[k];
  uint8_t flag = in[(size_t)i * stride + 96];
  AffineDev o;
  if (flag) {
    fe_zero(o.p.x);
    fe_zero(o.p.y);
  } else {
    fe_from_abi<F>(o.p.x, w, md);
    fe_from_abi<F>(o.p.y, w + 12, md);
    // canonical coordinates keep the "class M" contract tight and make equal points bit-identical
    fe_reduce<F>(o.p.x);
    fe_reduce<F>(o.p.y);
  }
  out[i] = o;
  inf[i] = flag ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// Fr Montgomery form (a * 2^256 mod r, what arkworks' `Fr` holds) -> the plain integer a: one Montgomery reduction
// over 8 x 32-bit limbs.  This is `into_bigint` of VariableBaseMSM::msm (ARK ec/src/msm/variable_base/mod.rs:48-53,
// ff montgomery_backend.rs:445-465) and sppark's `mont` flag (SPK msm/pippenger.cuh:157-164).
template <class FR>
__device__ __forceinline__ void fr_from_montgomery(uint32_t (&s)[8]) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t m = s[0] * FR::RINV;
    uint64_t c = ((uint64_t)m * FR::R[0] + s[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      c += (uint64_t)m * FR::R[j] + s[j];
      s[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    s[7] = (uint32_t)c;
  }
  // result < 2r; bring it below r
  uint32_t t[8];
  int64_t b = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    b += (int64_t)s[j] - FR::R[j];
    t[j] = (uint32_t)b;
    b >>= 32;
  }
  if (b == 0) {
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = t[j];
  }
}

// One thread per scalar.  Digits d_w in [-2^(c-1), 2^(c-1)] with sum d_w 2^(cw) = k; windows*c >= 257 so
// the last carry always lands in a window.  Zero digits (and every digit of a base flagged infinite)
// get the sentinel key, which sorts behind every real bucket.
template <class FR, bool MONT>
__global__ void __launch_bounds__(256) k_digits(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf,
                                                uint32_t n, uint32_t c, uint32_t windows,
                                                uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint4* sp = reinterpret_cast<const uint4*>(scalars) + 2 * (size_t)i;
  uint4 lo = sp[0], hi = sp[1];
  uint32_t s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  if (MONT) fr_from_montgomery<FR>(s);
  const bool dead = inf[i] != 0;
  const uint32_t half = 1u << (c - 1);
  const uint32_t sentinel = windows * half;
  const uint32_t wmask = (1u << c) - 1;
  uint32_t carry = 0;
  for (uint32_t w = 0; w < windows; w++) {
    uint32_t v = (s[0] & wmask) + carry;
#pragma unroll
    for (int j = 0; j < 7; j++) s[j] = (s[j] >> c) | (s[j + 1] << (32 - c));
    s[7] >>= c;
    const bool neg = v > half;
    const uint32_t d = neg ? (1u << c) - v : v;
    carry = neg ? 1u : 0u;
    const uint32_t key = (d == 0 || dead) ? sentinel : w * half + (d - 1);
    keys[(size_t)w * n + i] = key;
    vals[(size_t)w * n + i] = i | (neg ? 0x80000000u : 0u);
  }
}

// ------------------------------------------------------------------------------------------------
// Where a finished run fragment goes.  A fragment is the lane's partial sum for one key.  It is the
// whole bucket only if the run cannot continue into a neighbouring lane.
struct SegOut {
  XyzzDev* buckets;
  XyzzDev* slots;       // 2 per lane: [2t] head, [2t+1] tail
  uint32_t* slot_keys;  // KEY_NONE = empty slot
};

__device__ __forceinline__ void seg_flush(const SegOut& o, uint32_t t, uint32_t nlanes, uint32_t key, const Xyzz& acc,
                                          bool is_first, bool is_last) {
  const bool complete = (!is_first || t == 0) && (!is_last || t == nlanes - 1);
  XyzzDev v;
  v.p = acc;
  if (complete) {
    o.buckets[key] = v;
  } else if (is_first) {
    o.slots[2 * (size_t)t] = v;
    o.slot_keys[2 * (size_t)t] = key;
  } else {
    o.slots[2 * (size_t)t + 1] = v;
    o.slot_keys[2 * (size_t)t + 1] = key;
  }
}

// The hot kernel.  Lane t walks sorted entries [t*K, (t+1)*K): ~K mixed adds, one bucket store per run.
// The next base is fetched before the current add so the gather latency hides under ~5k VALU ops.
template <class F>
__global__ void __launch_bounds__(256, 2) k_accumulate(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                    uint32_t n_entries, uint32_t K, uint32_t sentinel,
                                                    const AffineDev* __restrict__ bases, SegOut out, uint32_t nlanes) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nlanes) return;
  Modulus<F> md;
  out.slot_keys[2 * (size_t)t] = KEY_NONE;
  out.slot_keys[2 * (size_t)t + 1] = KEY_NONE;
  const uint64_t beg = (uint64_t)t * K;
  const uint64_t end = (beg + K < n_entries) ? beg + K : n_entries;
  if (beg >= end) return;

  uint32_t key_n = keys[beg], val_n = vals[beg];
  AffineDev p_n;
  if (key_n != sentinel) p_n = bases[val_n & IDX_MASK];

  uint32_t cur = KEY_NONE;
  bool first = true, fresh = true;
  Xyzz acc;
  xyzz_set_inf<F>(acc);
  for (uint64_t e = beg; e < end; e++) {
    const uint32_t key = key_n, val = val_n;
    if (key == sentinel) break;  // sorted: nothing but sentinels from here on
    const Affine p = p_n.p;
    if (e + 1 < end) {
      key_n = keys[e + 1];
      val_n = vals[e + 1];
      if (key_n != sentinel) p_n = bases[val_n & IDX_MASK];
    }
    if (key != cur) {
      if (cur != KEY_NONE) {
        seg_flush(out, t, nlanes, cur, acc, first, false);
        first = false;
      }
      cur = key;
      fresh = true;
    }
    xyzz_madd<F>(acc, p, (val >> 31) != 0, fresh, md);
    fresh = false;
  }
  if (cur != KEY_NONE) seg_flush(out, t, nlanes, cur, acc, first, true);
}

// Merge run fragments: same walk over the slot sequence of the previous level (keys non-decreasing,
// KEY_NONE = hole), full XYZZ adds.  Recursion ends when one lane covers everything.
template <class F>
__global__ void __launch_bounds__(256) k_segreduce(const XyzzDev* __restrict__ in_slots, const uint32_t* __restrict__ in_keys,
                                                   uint32_t n_in, uint32_t K, SegOut out, uint32_t nlanes) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nlanes) return;
  Modulus<F> md;
  out.slot_keys[2 * (size_t)t] = KEY_NONE;
  out.slot_keys[2 * (size_t)t + 1] = KEY_NONE;
  const uint64_t beg = (uint64_t)t * K;
  const uint64_t end = (beg + K < n_in) ? beg + K : n_in;
  uint32_t cur = KEY_NONE;
  bool first = true;
  Xyzz acc;
  xyzz_set_inf<F>(acc);
  for (uint64_t e = beg; e < end; e++) {
    const uint32_t key = in_keys[e];
    if (key == KEY_NONE) continue;
    const XyzzDev v = in_slots[e];
    if (key != cur) {
      if (cur != KEY_NONE) {
        seg_flush(out, t, nlanes, cur, acc, first, false);
        first = false;
      }
      cur = key;
      acc = v.p;
    } else {
      xyzz_add<F>(acc, v.p, md);
    }
  }
  if (cur != KEY_NONE) seg_flush(out, t, nlanes, cur, acc, first, true);
}

// ------------------------------------------------------------------------------------------------
// Bucket -> window reduction, one level.  Per window the target is
//     V = sum_j A_j + sum_j weight(j) * X_j,    weight(j) = j + 1 on the first level, j afterwards,
// with X = buckets and no A on the first level.  Thread (w, t) owns chunk j in [tL, tL+L): it emits
//     A'_t = sum A_j + sum (local weight) X_j     and     X'_t = L * sum X_j        (L = 2^logL)
// so that V = sum_t A'_t + sum_t t * X'_t -- the same problem, L times smaller.  When one chunk is
// left, V = A'_0.  Running sums walk the chunk from the top: run += X_j; wsum += run.
template <class F, bool FIRST>
__global__ void __launch_bounds__(256) k_bucket_reduce(const XyzzDev* __restrict__ in_a, const XyzzDev* __restrict__ in_x,
                                                       uint32_t n_per_win, uint32_t logL, uint32_t chunks_per_win,
                                                       uint32_t windows, XyzzDev* __restrict__ out_a, XyzzDev* __restrict__ out_x) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= windows * chunks_per_win) return;
  Modulus<F> md;
  const uint32_t w = g / chunks_per_win, t = g % chunks_per_win;
  const uint32_t L = 1u << logL;
  const uint32_t lo = t * L;
  const uint32_t hi = (lo + L < n_per_win) ? lo + L : n_per_win;
  const XyzzDev* x = in_x + (size_t)w * n_per_win;
  Xyzz run, wsum;
  xyzz_set_inf<F>(run);
  xyzz_set_inf<F>(wsum);
  for (uint32_t j = hi; j-- > lo;) {
    const XyzzDev v = x[j];
    xyzz_add<F>(run, v.p, md);
    if (FIRST || j > lo) xyzz_add<F>(wsum, run, md);
  }
  if (!FIRST) {
    const XyzzDev* a = in_a + (size_t)w * n_per_win;
    for (uint32_t j = lo; j < hi; j++) {
      const XyzzDev v = a[j];
      xyzz_add<F>(wsum, v.p, md);
    }
  }
  if (!xyzz_is_inf<F>(run)) {
    for (uint32_t k = 0; k < logL; k++) xyzz_dbl<F>(run, md);
  }
  XyzzDev o;
  o.p = wsum;
  out_a[g] = o;
  o.p = run;
  out_x[g] = o;
}

}  // namespace msm
