// partition.hpp -- bucket grouping: scalars -> signed c-bit digits -> entries grouped by (window, bucket),
// as hand-written gfx950 kernels.  Replaces the library radix sort the first round used as scaffolding.
//
// What the reference does here: the CUB path of Matter Labs (ML msm.cu:229-330, "sorting is the bottleneck",
// P1A matter-labs/README.md:62-68) and yrrid's own partition + sort (CMB Partition1024.cu:157-243,
// Partition4096.cu:351-433, SortCounts.cu:54-185).  This is neither: an MSD partition whose FIRST level is fused with
// the digit extraction, so the W x N (key, value) pairs never exist as an unsorted 7-GB array that a sort has to re-read.
//
//   L1   tile = 8192 consecutive scalars, held in registers by a 1024-thread block for all W windows (32 B in, once).
//        k_l1_hist     digits of the tile -> per-tile histogram over the level-1 bins (window, top HB bits of the bucket),
//                      written as one row of a [tiles x bins] matrix (no atomics: deterministic offsets)
//        k_l1_scan_*   column scan of the matrix -> the offset of every (tile, bin) run, the bin (= segment) table
//        k_l1_scatter  digits again (cheaper than keeping 7 GB of them), LDS multisplit of the tile's 8192 entries of one
//                      window into its 512 or 1024 bins, flushed as ~16- or ~8-entry (128- / 64-B) contiguous runs of u64 entries
//                      (value = base index | sign << 31 in the low word, remaining bucket bits in the high word)
//   Pn   one or more generic passes over the segments the previous level produced, RB more bucket bits each:
//        k_pass_hist / k_pass_scan / k_pass_scatter -- a segment is cut into sub-jobs of <= SUBJOB entries, one block each, so
//        a skewed input (all scalars equal: one segment holds everything) is spread over the chip like a uniform one; offsets
//        again come from a (sub-job x bin) histogram matrix, not from atomics.  The last pass writes the full key into the
//        high word: the output is the (key, value) array the accumulation walks, fully sorted by key.
//
// Zero digits and bases flagged infinite produce NO entry (the count of real entries lives in device memory and the
// accumulation reads it), so there are no sentinel keys any more.
// With precomputed tables all windows share one bucket set: level-1 bins are then numbered bucket-major (top bits, then
// window), which makes the W per-window runs of a bucket range contiguous -- the same kernels, one more pass.
//
// Measured inputs to this design (tools/ubench_atomics.hip, profiles/r02_ubench_atomics.txt): device-scope atomics top out
// at 27 G/s (so per-tile claims by atomics would cost as much as the data movement), scattered stores need >= 64-B runs
// (16 B: 0.36 TB/s, 64 B: 3.2 TB/s, 128 B: 5.2 TB/s).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "digits.hpp"
#include "partition_plan.hpp"

namespace msm {

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`: every barrier of
// a scattering block then also waits for the global loads it issued ahead (the next tile) and for its global STORES to be
// acknowledged (gfx9 counts them in vmcnt) -- the memory pipeline drains at each of the five to eight barriers of a tile step, and
// the step costs compute + memory instead of their maximum (profiles/r04_ab_group_barrier.txt).  The kernels below exchange data
// between threads through LDS alone (global memory is read-only input or write-only output within a launch), so waiting for the
// LDS counter is all the barrier needs; the "memory" clobber keeps the compiler from moving accesses across it.
__device__ __forceinline__ void lds_barrier() {
#ifdef PART_FULL_BARRIER
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// ---- block-wide exclusive scan of one value per thread (blockDim.x a multiple of 64, <= 1024) --------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t u = __shfl_up(v, d, 64);
    if ((int)(threadIdx.x & 63) >= d) v += u;
  }
  return v;
}
// tmp: >= 17 words of LDS.  Returns the exclusive prefix of v; total of the block in `total`.  Ends with a barrier unless
// TAIL_BARRIER = false: the caller then guarantees a barrier of its own before `tmp` is used again.
template <bool TAIL_BARRIER = true>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* tmp, uint32_t& total) {
  const uint32_t incl = wave_incl_scan(v);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  if (lane == 63) tmp[wave] = incl;
  lds_barrier();
  if (wave == 0) {
    uint32_t w = lane < nw ? tmp[lane] : 0;
    const uint32_t wi = wave_incl_scan(w);
    if (lane < nw) tmp[lane] = wi - w;
    if (lane == nw - 1) tmp[16] = wi;
  }
  lds_barrier();
  const uint32_t r = tmp[wave] + incl - v;
  total = tmp[16];
  if (TAIL_BARRIER) lds_barrier();
  return r;
}

// ---- digits of one scalar, window after window --------------------------------------------------------------------------
struct ScalarDigits {
  uint32_t s[8];
  uint32_t carry;
};
// Next signed digit (CMB ProcessSignedDigits.cu:118-151 semantics): |d| in [0, 2^(c-1)], neg = the digit is negative.
// `flip` (the context option assume_subgroup): the top-bit trick of the ZPrize winners (CMB ProcessSignedDigits.cu:10-20,123-128:
// "if bit 252 of k is set use k' = r - k and -P"), here for every k in (r/2, r): k P = (r - k)(-P) whenever r P = O, and r - k has
// one significant bit less -- 252 = 12 x 21 bits for BLS12-377, so the window above them only ever receives the last carry (14.5 %
// of the scalars instead of 57 %).  r - k is never formed: its windows are rwin - (window of k) - borrow, with `rwin` the same
// window of r (wave-uniform: scalar registers), the borrow in bit 1 of `carry` -- the level-1 scatter holds eight scalars per
// thread in a 128-VGPR budget, and neither a pass over them nor a ninth register per scalar fits (both spilled).
// FOLD = false (the default path) compiles to the plain recoding: the option costs nothing when it is off.
//
// The window is cut straight out of the scalar (round 4): bits [o, o + 32) with o = w c are one v_alignbit of two neighbouring
// words, and WHICH two is wave-uniform (w and c are), so the choice is a scalar branch, not a select chain.  Before, the whole
// 8-word scalar was shifted down by c bits after every window: 8 quarter-rate shifts per digit, half of what the level-1
// kernels spent (profiles/r04_ab_group_digits.txt).
template <int L>
__device__ __forceinline__ uint32_t scalar_bits_at(const uint32_t (&s)[8], uint32_t sh) {
  if constexpr (L >= 8) {
    return 0;
  } else {
    const uint32_t hi = L < 7 ? s[L < 7 ? L + 1 : 7] : 0u;
    return __builtin_amdgcn_alignbit(hi, s[L], sh);
  }
}
__device__ __forceinline__ uint32_t scalar_window(const uint32_t (&s)[8], uint32_t o) {
  const uint32_t sh = o & 31;
  switch (o >> 5) {   // wave-uniform
    case 0: return scalar_bits_at<0>(s, sh);
    case 1: return scalar_bits_at<1>(s, sh);
    case 2: return scalar_bits_at<2>(s, sh);
    case 3: return scalar_bits_at<3>(s, sh);
    case 4: return scalar_bits_at<4>(s, sh);
    case 5: return scalar_bits_at<5>(s, sh);
    case 6: return scalar_bits_at<6>(s, sh);
    case 7: return scalar_bits_at<7>(s, sh);
    default: return 0;
  }
}
// the digit of window w: `u` = its c raw bits (scalar_window(st.s, w c) & wmask)
//
// `anchor` (wave-uniform: this window is PartPlan::anchor -- round 6, never together with FOLD): the carry chain ENDS here.  The window's
// value v = bits + carry in [0, 2^c] is written as 2^(c-1) + s with s in [-2^(c-1), 2^(c-1)]: |s| fits the same buckets, nothing is
// carried up, and the constant 2^(c-1) of EVERY scalar -- zero or not -- is a multiple of the plain sum of the bases that the engine
// adds on the host (msm_engine.hip, "anchored window").  The windows above start a fresh chain on the raw top bits.
template <bool FOLD>
__device__ __forceinline__ void next_digit(ScalarDigits& st, uint32_t u, uint32_t c, uint32_t half, uint32_t wmask, bool flip, uint32_t rwin,
                                           uint32_t& mag, bool& neg, bool anchor = false) {
  if constexpr (FOLD) {
    const uint32_t sub = u + ((st.carry >> 1) & 1u);   // <= 2^c
    const bool borrow = flip && rwin < sub;
    u = flip ? ((rwin - sub) & wmask) : u;
    const uint32_t v = u + (st.carry & 1u);
    const bool over = v > half;
    mag = over ? (1u << c) - v : v;
    neg = over != flip;
    st.carry = (over ? 1u : 0u) | (borrow ? 2u : 0u);
  } else {
    const uint32_t v = u + st.carry;
    if (anchor) {
      neg = v < half;
      mag = neg ? half - v : v - half;
      st.carry = 0;
    } else {
      neg = v > half;
      mag = neg ? (1u << c) - v : v;
      st.carry = neg ? 1u : 0u;
    }
  }
}

// window w of the scalar-field modulus (w, c wave-uniform)
template <class FR>
__device__ __forceinline__ uint32_t fr_window(uint32_t w, uint32_t c, uint32_t wmask) {
  const uint32_t o = w * c, limb = o >> 5, sh = o & 31;
  if (limb >= 8) return 0;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    if ((uint32_t)j == limb) lo = FR::R[j];
    if ((uint32_t)j == limb + 1) hi = FR::R[j];
  }
  return ((lo >> sh) | (sh ? hi << (32 - sh) : 0u)) & wmask;
}

// Folded or not, on the top limb alone: top(r)/2 < top(k) < top(r) implies r/2 < k < r; the 3e-9 of the scalars whose top limb
// EQUALS one of the bounds stay as they are, which is just as correct (folding is a per-scalar choice) and only means that their
// last carry may reach the window above.  Scalars >= r are left alone (any 256-bit integer stays legal).
template <class FR>
__device__ __forceinline__ bool fold_decision(const ScalarDigits& st) {
  return st.s[7] > (FR::R[7] >> 1) && st.s[7] < FR::R[7];
}

template <class FR, bool MONT>
__device__ __forceinline__ void load_scalar(ScalarDigits& st, const uint32_t* __restrict__ scalars, uint32_t i) {
  const uint4* sp = reinterpret_cast<const uint4*>(scalars) + 2 * (size_t)i;
  const uint4 lo = sp[0], hi = sp[1];
  uint32_t s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  if (MONT) fr_from_montgomery<FR>(s);
#pragma unroll
  for (int j = 0; j < 8; j++) st.s[j] = s[j];
  st.carry = 0;
}

// Tile of a level-1 block.  Workgroup b runs on XCD b % 8 (observed placement; used for speed only): give every XCD a CONTIGUOUS
// range of tiles, so that the tiles in flight on one XCD are neighbours.  Neighbouring tiles write neighbouring runs of every
// bin (the offsets are a column scan over tiles), and runs that share a 128-B line then meet in the same L2 before they are
// written back -- with the plain b -> tile map they would sit in eight different, non-coherent L2s as partial lines.
__device__ __forceinline__ uint32_t l1_tile(uint32_t block, uint32_t ntiles) {
  const uint32_t per_xcd = (ntiles + 7) >> 3;
  return (block & 7) * per_xcd + (block >> 3);
}

// Rank of an entry among the entries of its bin, and the bin's population, when there are only a FEW bins (b1 <= 8: small
// inputs, whose level 1 resolves few bits).  A plain LDS atomic per entry then serialises on a handful of addresses
// (8192 entries of a tile on ONE counter: 4 us per window, 0.5 ms for a 2^14 MSM); here the lanes of a wave that share a bin
// are counted with a ballot and ONE lane adds their number to the counter.
__device__ __forceinline__ uint32_t few_bins_rank(uint32_t* cnt, uint32_t b1, bool ok, uint32_t bin) {
  uint64_t mine = 0;
  for (uint32_t v = 0; v < b1; v++) {
    const uint64_t m = __builtin_amdgcn_ballot_w64(ok && bin == v);
    if (bin == v) mine = m;
  }
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t before = (uint32_t)__popcll(mine & ((1ull << lane) - 1));
  const int leader = mine ? __ffsll((unsigned long long)mine) - 1 : 0;
  uint32_t base = 0;
  if (ok && (int)lane == leader) base = atomicAdd(&cnt[bin], (uint32_t)__popcll(mine));
  base = __shfl(base, leader, 64);
  return base + before;
}
__device__ __forceinline__ void few_bins_count(uint32_t* cnt, uint32_t b1, uint32_t bin_base, bool ok, uint32_t bin) {
  for (uint32_t v = 0; v < b1; v++) {
    const uint64_t m = __builtin_amdgcn_ballot_w64(ok && bin == v);
    if (m && (threadIdx.x & 63) == (uint32_t)(__ffsll((unsigned long long)m) - 1)) atomicAdd(&cnt[bin_base + v], (uint32_t)__popcll(m));
  }
}

// Level-1 bin of (window w, top bucket bits hi): window-major normally.  With table levels the windows g, g + G, ... share bucket set
// g: their bins are NEIGHBOURS ((g b1 + hi) levels + j for level j = w / G), so that the runs of one bucket range form ONE segment
// for the next pass (k_l1_merge_shared).
__device__ __forceinline__ uint32_t l1_bin_of(const PartPlan& p, uint32_t w, uint32_t hi) {
  if (!p.shared) return w * p.b1 + hi;
  const uint32_t j = w / p.bsets, g = w - j * p.bsets;
  return (g * p.b1 + hi) * p.levels + j;
}
// the window a level-1 bin belongs to (>= windows for the unused bins of the last table level)
__device__ __forceinline__ uint32_t l1_window_of_bin(const PartPlan& p, uint32_t b) {
  if (!p.shared) return b / p.b1;
  const uint32_t q = b / p.levels, j = b - q * p.levels;
  return q / p.b1 + p.bsets * j;
}
// base index of scalar i in window w: table level w / bsets lies (w / bsets) * table_stride further on
__device__ __forceinline__ uint32_t l1_base_index(const PartPlan& p, uint32_t i, uint32_t w) {
  return p.idx0 + i + (p.table_stride ? (w / p.bsets) * p.table_stride : 0u);
}
// (bucket b): window-major normally; bucket-major with shared buckets, so that the W runs of one
// bucket range are neighbours and form ONE segment for the next pass.
__device__ __forceinline__ uint32_t l1_bin(const PartPlan& p, uint32_t w, uint32_t bucket) {
  const uint32_t hi = bucket >> p.lb;
  return l1_bin_of(p, w, hi);
}

// ------------------------------------------------------------------------------------------------------------------------
// L1 histogram: one block per tile, matrix row = the tile's entry count per level-1 bin.
template <class FR, bool MONT, bool FOLD>
__global__ void __launch_bounds__(PART_THREADS) k_l1_hist(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf, PartPlan p,
                                                          uint32_t* __restrict__ matrix) {
  extern __shared__ uint32_t hist[];   // nbins
  const uint32_t tile = l1_tile(blockIdx.x / p.wgroups, p.ntiles);
  if (tile >= p.ntiles) return;
  const uint32_t w_lo = (blockIdx.x % p.wgroups) * p.wper, w_hi = min(p.windows, w_lo + p.wper);   // this block's windows
  for (uint32_t b = threadIdx.x; b < p.nbins; b += PART_THREADS) hist[b] = 0;
  __syncthreads();
  const uint32_t wmask = (1u << p.c) - 1;
  const uint32_t i0 = tile * PART_TILE;
  const bool few = p.b1 <= 8;   // wave-uniform
#pragma unroll 1
  for (int k = 0; k < PART_PER_THREAD; k++) {
    const uint32_t i = i0 + threadIdx.x + k * PART_THREADS;
    const bool have = i < p.n;
    if (i0 + k * PART_THREADS >= p.n) break;   // (block-uniform) the tile ends here: a 1024-scalar MSM has one live slot of eight
    if (!few && !have) break;
    ScalarDigits st;
    bool flip = false;
    if (have) {
      load_scalar<FR, MONT>(st, scalars, i);
      if (FOLD) flip = fold_decision<FR>(st);
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) st.s[j] = 0;
      st.carry = 0;
    }
    const bool dead0 = !have || (p.table_stride ? false : inf[p.idx0 + i] != 0);
    for (uint32_t w = 0; w < p.windows; w++) {
      uint32_t mag;
      bool neg;
      next_digit<FOLD>(st, scalar_window(st.s, w * p.c) & wmask, p.c, p.half, wmask, flip, FOLD ? fr_window<FR>(w, p.c, wmask) : 0u, mag, neg, w == p.anchor);
      if (w < w_lo || w >= w_hi) continue;   // (block-uniform) another block of this tile counts that window
      bool dead = dead0;
      if (have && p.table_stride) dead = inf[l1_base_index(p, i, w)] != 0;
      const bool ok = mag != 0 && !dead;
      if (few) {
        // every lane of the wave takes part in the ballots (lanes past the end contribute nothing)
        const uint32_t hi = ok ? (mag - 1) >> p.lb : 0;
        if (p.shared) {
          for (uint32_t v = 0; v < p.b1; v++) {
            const uint64_t m = __builtin_amdgcn_ballot_w64(ok && hi == v);
            if (m && (threadIdx.x & 63) == (uint32_t)(__ffsll((unsigned long long)m) - 1)) atomicAdd(&hist[l1_bin_of(p, w, v)], (uint32_t)__popcll(m));
          }
        } else {
          few_bins_count(hist, p.b1, w * p.b1, ok, hi);
        }
      } else if (ok) {
#ifdef PART_EXP_L1H_NOATOMIC   // diagnosis only: the digits without the LDS atomics
        if (mag == 0x7fffffffu) hist[0] = 1;
#else
        atomicAdd(&hist[l1_bin(p, w, mag - 1)], 1u);
#endif
      }
    }
  }
  __syncthreads();
  // the row of a tile is written by its blocks together: each the bins of its own windows
  uint32_t* row = matrix + (size_t)tile * p.nbins;
  for (uint32_t b = threadIdx.x; b < p.nbins; b += PART_THREADS) {
    const uint32_t w = l1_window_of_bin(p, b);
    if ((w >= w_lo && w < w_hi) || (w >= p.windows && w_hi == p.windows)) row[b] = hist[b];   // (unused bins: zero, written by the last group)
  }
}

// Column scan of the [ntiles x nbins] matrix, three small kernels:
//   A  partial[g][b] = sum of the tiles of group g
//   B  (one block) exclusive scan over groups, then over bins: gbase[g][b] = first position of group g's entries of bin b;
//      writes the segment table (start, len, key_base), the sub-job prefix for the next pass and the entry total
//   C  in place: matrix[t][b] = position of tile t's run of bin b
__global__ void __launch_bounds__(256) k_l1_scan_a(const uint32_t* __restrict__ matrix, PartPlan p, uint32_t* __restrict__ partial) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (b >= p.nbins) return;
  const uint32_t t0 = g * p.tiles_per_group;
  const uint32_t t1 = min(p.ntiles, t0 + p.tiles_per_group);
  uint32_t sum = 0;
  for (uint32_t t = t0; t < t1; t++) sum += matrix[(size_t)t * p.nbins + b];
  partial[(size_t)g * p.nbins + b] = sum;
}

__global__ void __launch_bounds__(1024) k_l1_scan_b(uint32_t* __restrict__ partial, PartPlan p, PartSeg* __restrict__ segs,
                                                    uint32_t* __restrict__ subjob_first, uint32_t* __restrict__ totals) {
  __shared__ uint32_t tmp[32];
  __shared__ uint32_t carry_pos, carry_sj;
  if (threadIdx.x == 0) carry_pos = carry_sj = 0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < p.nbins; b0 += 1024) {
    const uint32_t b = b0 + threadIdx.x;
    // the 64 group sums of this bin: all loads in flight at once (a load-store-load chain on one array would serialise on
    // ~1 us of latency per step, 0.5 ms per launch), scanned in registers, written back below
    uint32_t gv[PART_SCAN_GROUPS];
    uint32_t col = 0;
    if (b < p.nbins) {
#pragma unroll
      for (int g = 0; g < PART_SCAN_GROUPS; g++) gv[g] = partial[(size_t)g * p.nbins + b];
#pragma unroll
      for (int g = 0; g < PART_SCAN_GROUPS; g++) {
        const uint32_t v = gv[g];
        gv[g] = col;   // exclusive over groups
        col += v;
      }
    }
    uint32_t tot, tot_sj;
    const uint32_t before = block_excl_scan(col, tmp, tot);
    const uint32_t nsj = part_subjobs_of(col);
    const uint32_t sj_before = block_excl_scan(nsj, tmp, tot_sj);
    const uint32_t base = carry_pos, base_sj = carry_sj;
    if (b < p.nbins) {
      const uint32_t start = base + before;
#pragma unroll
      for (int g = 0; g < PART_SCAN_GROUPS; g++) partial[(size_t)g * p.nbins + b] = gv[g] + start;
      // key bits resolved so far: window-major (w * half + hi << lb), or just hi << lb with shared buckets
      uint32_t key_base;
      if (p.shared)
        key_base = ((b / p.levels) / p.b1) * p.half + (((b / p.levels) % p.b1) << p.lb);   // bucket set g, top bits hi
      else
        key_base = (b / p.b1) * p.half + ((b % p.b1) << p.lb);
      segs[b] = PartSeg{start, col, key_base, 0};
      subjob_first[b] = base_sj + sj_before;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      carry_pos = base + tot;
      carry_sj = base_sj + tot_sj;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    subjob_first[p.nbins] = carry_sj;
    totals[0] = carry_pos;   // real entries
    totals[1] = carry_sj;    // sub-jobs of the next pass
  }
}

__global__ void __launch_bounds__(256) k_l1_scan_c(uint32_t* __restrict__ matrix, PartPlan p, const uint32_t* __restrict__ partial) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (b >= p.nbins) return;
  const uint32_t t0 = g * p.tiles_per_group;
  const uint32_t t1 = min(p.ntiles, t0 + p.tiles_per_group);
  uint32_t run = partial[(size_t)g * p.nbins + b];
  for (uint32_t t = t0; t < t1; t++) {
    const size_t at = (size_t)t * p.nbins + b;
    const uint32_t v = matrix[at];
    matrix[at] = run;
    run += v;
  }
}

// With table levels the segments of the next pass are the bsets * b1 bucket ranges, each the union of `levels` neighbouring bins.
__global__ void __launch_bounds__(256) k_l1_merge_shared(const PartSeg* __restrict__ bins, PartPlan p, PartSeg* __restrict__ segs,
                                                         uint32_t* __restrict__ subjob_first, uint32_t* __restrict__ totals) {
  // one block; bsets * b1 segments
  __shared__ uint32_t tmp[32];
  const uint32_t nseg = p.bsets * p.b1;
  for (uint32_t h0 = 0; h0 < nseg; h0 += 256) {
    const uint32_t h = h0 + threadIdx.x;
    uint32_t len = 0, start = 0;
    if (h < nseg) {
      start = bins[h * p.levels].start;
      for (uint32_t j = 0; j < p.levels; j++) len += bins[h * p.levels + j].len;
    }
    const uint32_t nsj = part_subjobs_of(len);
    uint32_t tot;
    const uint32_t before = block_excl_scan(nsj, tmp, tot);
    const uint32_t base = h0 == 0 ? 0 : subjob_first[h0];
    if (h < nseg) {
      segs[h] = PartSeg{start, len, (h / p.b1) * p.half + ((h % p.b1) << p.lb), 0};
      subjob_first[h] = base + before;
    }
    __syncthreads();
    if (threadIdx.x == 0) subjob_first[min(h0 + 256, nseg)] = base + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[1] = subjob_first[nseg];
}

// L1 scatter: one block per tile; the tile's scalars stay in registers while the windows are processed one after the other.
// LDS: stage (8192 x u64; the bin rides in the high half of the key word until the entry leaves) + three (b1 + 1)-word arrays.
template <class FR, bool MONT, bool FOLD>
__global__ void __launch_bounds__(PART_THREADS) k_l1_scatter(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf, PartPlan p,
                                                             const uint32_t* __restrict__ matrix, uint2* __restrict__ out) {
  __shared__ uint2 stage[PART_TILE];
  __shared__ uint32_t offs[1 << PART_MAX_HB], cnt[1 << PART_MAX_HB], tstart[(1 << PART_MAX_HB) + 1];
  __shared__ uint32_t tmp[32];
  const uint32_t tile = l1_tile(blockIdx.x / p.wgroups, p.ntiles);
  if (tile >= p.ntiles) return;
  const uint32_t w_lo = (blockIdx.x % p.wgroups) * p.wper, w_hi = min(p.windows, w_lo + p.wper);   // this block's windows
  const uint32_t wmask = (1u << p.c) - 1, lowmask = (1u << p.lb) - 1;
  const bool few = p.b1 <= 8;   // wave-uniform
  const uint32_t i0 = tile * PART_TILE;
  // slots of the tile that hold scalars at all (block-uniform): the steps below skip the others -- ballots and digit shifts
  // for seven empty slots were 3/4 of this kernel's time on a 1024-scalar MSM
  const uint32_t kmax = min((uint32_t)PART_PER_THREAD, (p.n - i0 + PART_THREADS - 1) / PART_THREADS);
  ScalarDigits st[PART_PER_THREAD];
  uint32_t alive = 0;     // bit k: scalar k exists and (without tables) its base is not flagged infinite; bit 8 + k: scalar k was folded
#pragma unroll
  for (int k = 0; k < PART_PER_THREAD; k++) {
    const uint32_t i = i0 + threadIdx.x + k * PART_THREADS;
    if (i < p.n) {
      load_scalar<FR, MONT>(st[k], scalars, i);
      if (p.table_stride || inf[p.idx0 + i] == 0) alive |= 1u << k;   // (with tables the flag is read per level, below)
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) st[k].s[j] = 0;
      st[k].carry = 0;
    }
  }
  if (FOLD) {
#pragma unroll
    for (int k = 0; k < PART_PER_THREAD; k++)
      if (fold_decision<FR>(st[k])) alive |= 0x100u << k;   // (an all-zero slot past the end is never folded)
  }
  const uint32_t* row = matrix + (size_t)tile * p.nbins;
  // the run offsets of the next window are fetched a whole window step ahead (a global load in the step's critical path
  // would cost ~1.5 us of the ~10 us a step takes)
  // windows below this block's range only advance the digit state (their carry feeds ours)
#pragma unroll
  for (int k = 0; k < PART_PER_THREAD; k++) {
    if ((uint32_t)k < kmax) {   // (block-uniform; slot by slot, so that empty slots cost nothing)
      for (uint32_t w = 0; w < w_lo; w++) {
        uint32_t mag;
        bool neg;
        next_digit<FOLD>(st[k], scalar_window(st[k].s, w * p.c) & wmask, p.c, p.half, wmask, ((alive >> (8 + k)) & 1) != 0,
                         FOLD ? fr_window<FR>(w, p.c, wmask) : 0u, mag, neg, w == p.anchor);
      }
    }
  }
  uint32_t offs_next = threadIdx.x < p.b1 ? row[l1_bin_of(p, w_lo, threadIdx.x)] : 0;
  if (threadIdx.x < p.b1) cnt[threadIdx.x] = 0;
  lds_barrier();
  // Barriers per window: after the ranking (A), two inside the scan, after the scan step (B), after staging (C).  cnt is zeroed and
  // the run offsets are installed IN the scan step; no barrier ends a window: the next window's A orders its scan step (which
  // rewrites tstart / offs) and its staging behind this window's copy-out.  (Seven barriers per window before.)
  for (uint32_t w = w_lo; w < w_hi; w++) {
    uint32_t where[PART_PER_THREAD];   // bin << 16 | rank in the tile's bin; 0xffffffff = no entry
    uint32_t low[PART_PER_THREAD];     // the bucket bits below the bin | sign << 31 (the entry's base index is recomputed when it is staged:
                                       // two registers per entry, not three -- this kernel runs at the 128-VGPR limit of a 1024-thread block;
                                       // same speed as three, same-box A/B)
    const uint32_t rwin = FOLD ? fr_window<FR>(w, p.c, wmask) : 0u;
    uint32_t raw[PART_PER_THREAD];    // the window's bits of the eight scalars: ONE uniform branch on the word pair, eight v_alignbit
    {
      const uint32_t o = w * p.c, sh = o & 31;
#define PART_RAW(L) case L: _Pragma("unroll") for (int k = 0; k < PART_PER_THREAD; k++) raw[k] = scalar_bits_at<L>(st[k].s, sh); break;
      switch (o >> 5) {
        PART_RAW(0) PART_RAW(1) PART_RAW(2) PART_RAW(3) PART_RAW(4) PART_RAW(5) PART_RAW(6) PART_RAW(7)
        default:
#pragma unroll
          for (int k = 0; k < PART_PER_THREAD; k++) raw[k] = 0;
      }
#undef PART_RAW
    }
#pragma unroll
    for (int k = 0; k < PART_PER_THREAD; k++) {
      if ((uint32_t)k >= kmax) {
        where[k] = 0xffffffffu;
        continue;
      }
      uint32_t mag;
      bool neg;
      next_digit<FOLD>(st[k], raw[k] & wmask, p.c, p.half, wmask, ((alive >> (8 + k)) & 1) != 0, rwin, mag, neg, w == p.anchor);
      const uint32_t i = i0 + threadIdx.x + k * PART_THREADS;
      bool ok = ((alive >> k) & 1) && mag != 0;
      const uint32_t idx = l1_base_index(p, i, w);
      if (ok && p.table_stride) ok = inf[idx] == 0;
      where[k] = 0xffffffffu;
      const uint32_t bucket = ok ? mag - 1 : 0, hi = bucket >> p.lb;
      uint32_t rank = 0;
      if (few)
        rank = few_bins_rank(cnt, p.b1, ok, hi);      // all lanes: ballots inside
      else if (ok)
        rank = atomicAdd(&cnt[hi], 1u);
      if (ok) {
        where[k] = (hi << 16) | rank;
        low[k] = (bucket & lowmask) | (neg ? 0x80000000u : 0u);   // lb <= 15
      }
    }
    lds_barrier();   // A
    {
      // exclusive scan of the tile's bin counts (b1 <= 1024 = blockDim)
      const uint32_t v = threadIdx.x < p.b1 ? cnt[threadIdx.x] : 0;
      uint32_t tot;
      const uint32_t ex = block_excl_scan<false>(v, tmp, tot);
      if (threadIdx.x < p.b1) {
        tstart[threadIdx.x] = ex;
        offs[threadIdx.x] = offs_next - ex;   // global position of staged slot j of this bin = offs[bin] + j: one LDS read per entry on the way out
        cnt[threadIdx.x] = 0;
        if (w + 1 < w_hi) offs_next = row[l1_bin_of(p, w + 1, threadIdx.x)];
      }
      if (threadIdx.x == 0) tstart[p.b1] = tot;
    }
    lds_barrier();   // B
#pragma unroll
    for (int k = 0; k < PART_PER_THREAD; k++)
      if (where[k] != 0xffffffffu) {
        const uint32_t idx = l1_base_index(p, i0 + threadIdx.x + k * PART_THREADS, w);
        // (value = base index | sign << 31; key word while staged: the bits still unresolved with the bin above them)
        stage[tstart[where[k] >> 16] + (where[k] & 0xffffu)] = make_uint2(idx | (low[k] & 0x80000000u), (low[k] & 0xffffu) | (where[k] & 0xffff0000u));
      }
    lds_barrier();   // C
    const uint32_t total = tstart[p.b1];
    for (uint32_t j = threadIdx.x; j < total; j += PART_THREADS) {
      uint2 e = stage[j];
      const uint32_t hi = e.y >> 16;
      e.y &= 0xffffu;
#ifdef PART_EXP_L1_NOSTORE   // diagnosis only (tools/group_probe.sh): everything but the global store
      if (e.x == 0x12345678u && e.y == 0x9abcu) out[0] = e;
#elif defined(PART_EXP_L1_OUT4)   // diagnosis only (wrong output; profiles/r05_ab_group_compact.txt): 4-byte entries between level 1 and the pass --
      reinterpret_cast<uint32_t*>(out)[offs[hi] + j] = e.x;   // the floor of ANY compact intermediate format (yrrid's is 5 bytes, CMB Partition1024.cu:157-243)
#else
      out[offs[hi] + j] = e;
#endif
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Generic pass over the segments of the previous level: RB more bits (the TOP rb of the `rem` bits still in the high word).

// sub-job j -> (segment, entry range)
__device__ __forceinline__ bool subjob_range(const PartSeg* __restrict__ segs, const uint32_t* __restrict__ subjob_first, uint32_t nsegs,
                                             uint32_t j, uint32_t& seg, uint32_t& beg, uint32_t& end) {
  if (j >= subjob_first[nsegs]) return false;
  uint32_t lo = 0, hi = nsegs;   // last seg with subjob_first[seg] <= j (segments without sub-jobs share their successor's value)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (subjob_first[mid] <= j) lo = mid; else hi = mid;
  }
  // skip empty segments that start at the same sub-job index
  while (lo + 1 < nsegs && subjob_first[lo + 1] <= j) lo++;
  seg = lo;
  const PartSeg s = segs[seg];
  const uint32_t k = j - subjob_first[seg];
  beg = s.start + k * PART_SUBJOB;
  end = min(s.start + s.len, beg + PART_SUBJOB);
  return beg < end;
}

// histogram row of sub-job j: counts[j][2^rb]
__global__ void __launch_bounds__(1024) k_pass_hist(const uint2* __restrict__ in, const PartSeg* __restrict__ segs,
                                                    const uint32_t* __restrict__ subjob_first, PassPlan pp, uint32_t* __restrict__ counts) {
  __shared__ uint32_t hist[1 << PART_MAX_RB];
  const uint32_t nb = 1u << pp.rb, shift = pp.rem - pp.rb;
  // the blocks walk the sub-jobs (none at all for a uniform input, whose segments all go to k_pass_fused: the grid is small so that
  // finding nothing to do costs microseconds, not a block launch per possible sub-job)
  const uint32_t nsub = subjob_first[pp.nsegs];
  for (uint32_t j = blockIdx.x; j < nsub; j += gridDim.x) {
    uint32_t seg, beg, end;
    const bool live = subjob_range(segs, subjob_first, pp.nsegs, j, seg, beg, end);
    if (!live || part_fused_takes(segs[seg].len)) continue;   // (block-uniform)
    for (uint32_t b = threadIdx.x; b < nb; b += 1024) hist[b] = 0;
    __syncthreads();
    for (uint32_t e = beg + threadIdx.x; e < end; e += 1024) atomicAdd(&hist[(in[e].y >> shift) & (nb - 1)], 1u);
    __syncthreads();
    uint32_t* row = counts + (size_t)j * nb;
    for (uint32_t b = threadIdx.x; b < nb; b += 1024) row[b] = hist[b];
    __syncthreads();
  }
}

// One block per segment: turns the count rows of its sub-jobs into positions (in place) and emits the sub-segments.
// The output range of a segment is its input range (the partition is in place segment by segment, between two buffers).
__global__ void __launch_bounds__(1024) k_pass_scan(const PartSeg* __restrict__ segs, const uint32_t* __restrict__ subjob_first, PassPlan pp,
                                                    uint32_t* __restrict__ counts, PartSeg* __restrict__ out_segs) {
  __shared__ uint32_t tmp[32];
  const uint32_t s = blockIdx.x, nb = 1u << pp.rb, b = threadIdx.x;
  const PartSeg sg = segs[s];
  if (part_fused_takes(sg.len)) return;
  const uint32_t j0 = subjob_first[s], j1 = subjob_first[s + 1];
  uint32_t tot_b = 0;
  if (b < nb)
    for (uint32_t j = j0; j < j1; j++) tot_b += counts[(size_t)j * nb + b];
  uint32_t tot;
  const uint32_t before = block_excl_scan(tot_b, tmp, tot);
  if (b < nb) {
    uint32_t run = sg.start + before;
    if (out_segs) out_segs[(size_t)s * nb + b] = PartSeg{run, tot_b, sg.key_base + (b << (pp.rem - pp.rb)), 0};
    for (uint32_t j = j0; j < j1; j++) {
      const size_t at = (size_t)j * nb + b;
      const uint32_t v = counts[at];
      counts[at] = run;
      run += v;
    }
  }
}

// sub-job prefix for the NEXT pass over the sub-segments this pass produced (one block; nsegs_out <= a few million)
__global__ void __launch_bounds__(1024) k_pass_subjobs(const PartSeg* __restrict__ segs, uint32_t nsegs, uint32_t* __restrict__ subjob_first,
                                                       uint32_t* __restrict__ totals) {
  __shared__ uint32_t tmp[32];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t s0 = 0; s0 < nsegs; s0 += 1024) {
    const uint32_t s = s0 + threadIdx.x;
    const uint32_t nsj = s < nsegs ? part_subjobs_of(segs[s].len) : 0;
    uint32_t tot;
    const uint32_t before = block_excl_scan(nsj, tmp, tot);
    const uint32_t base = carry;
    if (s < nsegs) subjob_first[s] = base + before;
    __syncthreads();
    if (threadIdx.x == 0) carry = base + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    subjob_first[nsegs] = carry;
    totals[1] = carry;
  }
}

// Scatter of one sub-job: LDS multisplit tile by tile, cursors of the sub-job in LDS, runs written contiguously; the next
// tile's entries are already in flight while the current one is split.  What bounds it is not LDS or issue but how many
// partially written 128-B lines the XCD's 4-MB L2 has to keep open (bins x concurrent blocks): hence big tiles, one block per CU.
#ifdef PART_EXP_PS_CACHEDLOAD   // diagnosis only: every block re-reads one 128-KB window (L2 hits)
#define PART_LD(e) ((e) & 0x3fffu)
#else
#define PART_LD(e) (e)
#endif
// diagnosis only (wrong output): the pass READS 4-byte entries; the bits it partitions by are taken from the value so that the bins stay
// uniformly filled
#ifdef PART_EXP_PS_IN4
__device__ __forceinline__ uint2 part_in(const uint2* __restrict__ in, uint32_t e) {
  const uint32_t v = reinterpret_cast<const uint32_t*>(in)[e];
  return make_uint2(v, (v * 2654435761u) >> 16);
}
#define PART_IN(in, e) part_in(in, e)
#else
#define PART_IN(in, e) ((in)[e])
#endif
// LDS of a scattering block: the staged tile, and per bin the cursor of the (sub-)job, the tile's count / first staged slot /
// global position of its staged slot 0.
struct PassLds {
  uint2 stage[PART_PTILE];
  uint32_t cur[1 << PART_MAX_RB], cnt[1 << PART_MAX_RB], tstart[(1 << PART_MAX_RB) + 1], dst[1 << PART_MAX_RB];
  uint32_t tmp[32];
};

// Scatter of the entries [beg, end) of one segment; the caller has set the cursors L.cur, zeroed L.cnt and passed a barrier:
// LDS multisplit tile by tile, runs written contiguously; the next tile's entries are already in flight while the current one
// is split.
__device__ __forceinline__ void pass_scatter_tiles(PassLds& L, const uint2* __restrict__ in, uint2* __restrict__ out, uint32_t beg, uint32_t end,
                                                   const PassPlan& pp, uint32_t key_base) {
  const uint32_t nb = 1u << pp.rb, shift = pp.rem - pp.rb, keepmask = (1u << shift) - 1;
  constexpr int PER = PART_PTILE / 1024;
  uint2 nxt[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const uint32_t e = beg + threadIdx.x + k * 1024;
    if (e < end) nxt[k] = PART_IN(in, PART_LD(e));
  }
  // Barriers per tile: after the ranking (A), two inside the scan, after the scan step (B), after staging (C).  L.cnt is zeroed and
  // L.cur advanced IN the scan step, and there is no barrier at the end of a tile: the next tile's A orders its scan step (which
  // rewrites tstart / dst) and its staging behind this tile's copy-out.  (Eight barriers per tile before; each one stalls 16 waves.)
  for (uint32_t t0 = beg; t0 < end; t0 += PART_PTILE) {
    uint2 ent[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
      ent[k] = nxt[k];
      const uint32_t e = t0 + PART_PTILE + threadIdx.x + k * 1024;
      if (e < end) nxt[k] = PART_IN(in, PART_LD(e));
    }
    uint32_t where[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const uint32_t e = t0 + threadIdx.x + k * 1024;
      where[k] = 0xffffffffu;
      if (e < end) {
        const uint32_t bin = (ent[k].y >> shift) & (nb - 1);
        const uint32_t rank = atomicAdd(&L.cnt[bin], 1u);
        where[k] = (bin << 16) | rank;
        // key word while staged: the bits still unresolved (< 2^16) with the bin above them
        ent[k].y = (ent[k].y & keepmask) | (bin << 16);
      }
    }
    lds_barrier();   // A
    {
      const uint32_t v = threadIdx.x < nb ? L.cnt[threadIdx.x] : 0;
      uint32_t tot;
      const uint32_t ex = block_excl_scan<false>(v, L.tmp, tot);
      if (threadIdx.x < nb) {
        const uint32_t c0 = L.cur[threadIdx.x];
        L.tstart[threadIdx.x] = ex;
#ifdef PART_EXP_PS_ALIGNED   // diagnosis only (wrong output): every run starts on a 128-byte line
        L.dst[threadIdx.x] = (c0 & ~15u) - ex;
#else
        L.dst[threadIdx.x] = c0 - ex;   // global position of staged slot j of this bin = dst[bin] + j
#endif
        L.cur[threadIdx.x] = c0 + v;
        L.cnt[threadIdx.x] = 0;
      }
      if (threadIdx.x == 0) L.tstart[nb] = tot;
    }
    lds_barrier();   // B
#pragma unroll
    for (int k = 0; k < PER; k++)
      if (where[k] != 0xffffffffu) L.stage[L.tstart[where[k] >> 16] + (where[k] & 0xffffu)] = ent[k];
    lds_barrier();   // C
    const uint32_t total = L.tstart[nb];
    for (uint32_t j = threadIdx.x; j < total; j += 1024) {
      uint2 e = L.stage[j];
      const uint32_t bin = e.y >> 16;
      // leaving: after the last pass the key word is the full key, before it the bits the next pass will look at
      e.y = pp.last ? key_base + bin : (e.y & 0xffffu);
#if defined(PART_EXP_PS_NOSTORE)
      if (e.x == 0x12345678u && e.y == 0x9abcu) out[0] = e;
#elif defined(PART_EXP_PS_L2STORE)   // diagnosis only: the stores land in a 1-MB window (L2 hits, no HBM write traffic)
      out[(L.dst[bin] + j) & 0x1ffffu] = e;
#elif defined(PART_EXP_PS_OUT4)      // diagnosis only (wrong output): the sorted output as 4-byte values (the key would come from a per-bucket offset table)
      reinterpret_cast<uint32_t*>(out)[L.dst[bin] + j] = e.x;
#else
      out[L.dst[bin] + j] = e;
#endif
    }
  }
}

// The scatter of a generic pass, one launch for both kinds of segment.
//
// Blocks [gen, gen + nsegs): one block per segment of at most PART_SUBJOB entries -- every segment of a uniform input: histogram of
// the segment, scan, the sub-segments for the next pass, and the scatter, in one kernel.  The histogram sweep streams the segment
// from HBM; the scatter sweep finds most of it again in L2 / the memory-side cache (512 KB per segment at 2^26 pairs), so the
// entries cross the HBM interface once on the way in instead of twice, and no (sub-job x bin) matrix is written for them (round 4).
//
// Blocks [0, gen): the sub-jobs of the LONGER segments, whose positions k_pass_hist + k_pass_scan prepared (a skewed input: all
// scalars equal puts a whole window into one segment; and the top window of any input, whose few significant bits fill only a
// handful of level-1 bins -- 344 sub-jobs at 2^26 pairs).  They walk the sub-jobs, and they come FIRST in the grid, so that these
// few long jobs run under the thousands of short ones instead of leaving most of the chip idle in a launch of their own.
//
// What bounds the scatter is not LDS or issue but how many partially written 128-B lines the XCD's 4-MB L2 has to keep open
// (bins x concurrent blocks): hence big tiles, one block per CU.
__global__ void __launch_bounds__(1024, PART_PASS_OCC) k_pass_scatter(const uint2* __restrict__ in, const PartSeg* __restrict__ segs,
                                                          const uint32_t* __restrict__ subjob_first, PassPlan pp, uint32_t gen,
                                                          const uint32_t* __restrict__ positions, uint2* __restrict__ out,
                                                          PartSeg* __restrict__ out_segs) {
  __shared__ PassLds L;
  const uint32_t nb = 1u << pp.rb, shift = pp.rem - pp.rb;
  if (blockIdx.x < gen) {
    const uint32_t nsub = subjob_first[pp.nsegs];
    for (uint32_t j = blockIdx.x; j < nsub; j += gen) {
      uint32_t seg, beg, end;
      const bool live = subjob_range(segs, subjob_first, pp.nsegs, j, seg, beg, end);
      if (!live) continue;
      const PartSeg sg = segs[seg];
      if (part_fused_takes(sg.len)) continue;
      const uint32_t* row = positions + (size_t)j * nb;
      lds_barrier();   // the previous sub-job's copy-out still reads L
      for (uint32_t b = threadIdx.x; b < nb; b += 1024) {
        L.cur[b] = row[b];
        L.cnt[b] = 0;
      }
      lds_barrier();
      pass_scatter_tiles(L, in, out, beg, end, pp, sg.key_base);
    }
    return;
  }
  const uint32_t s = blockIdx.x - gen;
  const PartSeg sg = segs[s];
  if (!part_fused_takes(sg.len)) return;
  if (sg.len == 0) {
    if (out_segs)
      for (uint32_t b = threadIdx.x; b < nb; b += 1024) out_segs[(size_t)s * nb + b] = PartSeg{sg.start, 0, sg.key_base + (b << shift), 0};
    return;
  }
  const uint32_t beg = sg.start, end = sg.start + sg.len;
  for (uint32_t b = threadIdx.x; b < nb; b += 1024) L.cnt[b] = 0;
  lds_barrier();
  {
    // four loads in flight per thread
    uint32_t e = beg + threadIdx.x;
    for (; e + 3 * 1024 < end; e += 4 * 1024) {
      const uint32_t k0 = PART_IN(in, e).y, k1 = PART_IN(in, e + 1024).y, k2 = PART_IN(in, e + 2048).y, k3 = PART_IN(in, e + 3072).y;
      atomicAdd(&L.cnt[(k0 >> shift) & (nb - 1)], 1u);
      atomicAdd(&L.cnt[(k1 >> shift) & (nb - 1)], 1u);
      atomicAdd(&L.cnt[(k2 >> shift) & (nb - 1)], 1u);
      atomicAdd(&L.cnt[(k3 >> shift) & (nb - 1)], 1u);
    }
    for (; e < end; e += 1024) atomicAdd(&L.cnt[(PART_IN(in, e).y >> shift) & (nb - 1)], 1u);
  }
  lds_barrier();
  {
    const uint32_t v = threadIdx.x < nb ? L.cnt[threadIdx.x] : 0;
    uint32_t tot;
    const uint32_t ex = block_excl_scan(v, L.tmp, tot);
    if (threadIdx.x < nb) {
      L.cur[threadIdx.x] = beg + ex;
      L.cnt[threadIdx.x] = 0;
      if (out_segs) out_segs[(size_t)s * nb + threadIdx.x] = PartSeg{beg + ex, v, sg.key_base + (threadIdx.x << shift), 0};
    }
  }
  lds_barrier();
  pass_scatter_tiles(L, in, out, beg, end, pp, sg.key_base);
}

}  // namespace msm

// ---- the launch sequence -------------------------------------------------------------------------------------------------------
namespace msm {

#ifdef MSM_DEBUG
// ---- invariant checks of the debug build (-DMSM_DEBUG: tools/build_debug.sh, tests/test_gpu_debug_build.py) -----------------------
// The reference keeps a (disabled) prefix / count self-check in its partition ("super useful debugging code for catching race
// conditions", CMB Partition4096.cu:419-432).  Here: after level 1 and after every pass the segment table must be contiguous and
// end at the entry total, and every entry of a segment must carry only the key bits still unresolved; after the last pass the keys
// must be non-decreasing, below the key limit, and the values must name bases of this chunk; the number of entries must equal the
// number of non-zero digits, counted independently by shifting the whole scalar down window by window (NOT by scalar_window);
// after the accumulation every slot key must be KEY_NONE or a valid key.  Violations are COUNTED into totals[4 + i]:
enum { DBG_DIGITS = 0, DBG_SEG_GAP = 1, DBG_SEG_END = 2, DBG_KEY_BITS = 3, DBG_UNSORTED = 4, DBG_KEY_RANGE = 5, DBG_VALUE_RANGE = 6,
       DBG_SLOT_KEY = 7, DBG_CHECKS = 8, DBG_WORDS = 12 };

template <class FR, bool MONT, bool FOLD>
__global__ void __launch_bounds__(256) k_dbg_count_digits(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf, PartPlan p,
                                                          uint32_t* __restrict__ dbg) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  uint32_t count = 0;
  if (i < p.n) {
    ScalarDigits st;
    load_scalar<FR, MONT>(st, scalars, i);
    const bool flip = FOLD ? fold_decision<FR>(st) : false;
    const uint32_t wmask = (1u << p.c) - 1;
    for (uint32_t w = 0; w < p.windows; w++) {
      const uint32_t u = st.s[0] & wmask;                 // the plain way: low bits, then the whole scalar down by c
      for (int j = 0; j < 7; j++) st.s[j] = (st.s[j] >> p.c) | (st.s[j + 1] << (32 - p.c));
      st.s[7] >>= p.c;
      uint32_t mag;
      bool neg;
      next_digit<FOLD>(st, u, p.c, p.half, wmask, flip, FOLD ? fr_window<FR>(w, p.c, wmask) : 0u, mag, neg, w == p.anchor);
      const bool dead = inf[l1_base_index(p, i, w)] != 0;
      if (mag != 0 && !dead) count++;
    }
  }
  for (int d = 32; d > 0; d >>= 1) count += __shfl_down(count, d, 64);
  if ((threadIdx.x & 63) == 0 && count) atomicAdd(&dbg[DBG_DIGITS], count);
}

// one block per segment: contiguity of the table, and only `rem` key bits left in its entries
__global__ void __launch_bounds__(256) k_dbg_check_level(const uint2* __restrict__ entries, const PartSeg* __restrict__ segs, uint32_t nsegs, uint32_t rem,
                                                         const uint32_t* __restrict__ totals, uint32_t* __restrict__ dbg) {
  const uint32_t s = blockIdx.x;
  const PartSeg sg = segs[s];
  if (threadIdx.x == 0) {
    const uint32_t expect = s ? segs[s - 1].start + segs[s - 1].len : 0u;
    if (sg.start != expect) atomicAdd(&dbg[DBG_SEG_GAP], 1u);
    if (s + 1 == nsegs && sg.start + sg.len != totals[0]) atomicAdd(&dbg[DBG_SEG_END], 1u);
  }
  uint32_t bad = 0;
  for (uint32_t e = sg.start + threadIdx.x; e < sg.start + sg.len; e += 256)
    if (rem < 32 && (entries[e].y >> rem) != 0) bad++;
  if (bad) atomicAdd(&dbg[DBG_KEY_BITS], bad);
}

__global__ void __launch_bounds__(256) k_dbg_check_sorted(const uint2* __restrict__ entries, const uint32_t* __restrict__ totals, uint32_t key_limit,
                                                          uint32_t idx_lo, uint32_t idx_hi, uint32_t* __restrict__ dbg) {
  const uint32_t n = totals[0];
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint2 e = entries[i];
    if (i + 1 < n && entries[i + 1].y < e.y) atomicAdd(&dbg[DBG_UNSORTED], 1u);
    if (e.y >= key_limit) atomicAdd(&dbg[DBG_KEY_RANGE], 1u);
    const uint32_t idx = e.x & 0x7fffffffu;
    if (idx < idx_lo || idx >= idx_hi) atomicAdd(&dbg[DBG_VALUE_RANGE], 1u);
  }
}

__global__ void __launch_bounds__(256) k_dbg_check_slots(const uint32_t* __restrict__ keys, uint32_t n, uint32_t key_limit, uint32_t* __restrict__ dbg) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && keys[i] != 0xffffffffu && keys[i] >= key_limit) atomicAdd(&dbg[DBG_SLOT_KEY], 1u);
  if (i == 0) atomicAdd(&dbg[DBG_CHECKS], 1u);
}

// test hook of the debug build: swap the keys of two neighbouring entries of different keys (MI355_MSM_DEBUG_CORRUPT=1), so that a
// test can see the checks FAIL
__global__ void k_dbg_corrupt(uint2* __restrict__ entries, const uint32_t* __restrict__ totals) {
  const uint32_t n = totals[0];
  for (uint32_t i = 0; i + 1 < n; i++)
    if (entries[i].y != entries[i + 1].y) {
      const uint32_t k = entries[i].y;
      entries[i].y = entries[i + 1].y;
      entries[i + 1].y = k;
      return;
    }
}
#endif   // MSM_DEBUG


#ifndef PART_GEN_GRID
#define PART_GEN_GRID 512u
#endif
// Optional per-kernel timing of one grouping run (tools/partition_test.hip): an event after every launch.
struct PartProbe {
  static constexpr int MAX = 32;
  hipEvent_t ev[MAX + 1];
  const char* name[MAX];
  int n = 0;
  void mark(const char* what, hipStream_t st) {
    if (n < MAX) {
      name[n] = what;
      (void)hipEventRecord(ev[++n], st);
    }
  }
};
#define PART_MARK(what) do { if (probe) probe->mark(what, st); } while (0)

// Enqueues the whole grouping on `st`; returns the index (0/1) of the entry buffer that holds the result.
// `mid` (optional) is recorded after level 1 so the caller can time the two halves.
template <class FR, bool MONT>
inline int part_run(const uint32_t* d_scalars, const uint8_t* d_inf, const PartPlan& p, const PartBuffers& b, hipStream_t st,
                    hipEvent_t mid, hipError_t& err, PartProbe* probe = nullptr) {
  err = hipSuccess;
  if (probe) {
    probe->n = 0;
    (void)hipEventRecord(probe->ev[0], st);
  }
#ifdef MSM_DEBUG
  uint32_t* const dbg = b.totals + 4;
  (void)hipMemsetAsync(dbg, 0, DBG_WORDS * 4, st);
#define PART_DBG_LEVEL(ENT, SEGS, NSEGS, REM)                                                                              \
  do {                                                                                                                     \
    hipLaunchKernelGGL(k_dbg_check_level, dim3((uint32_t)(NSEGS)), dim3(256), 0, st, ENT, SEGS, (uint32_t)(NSEGS), REM, b.totals, dbg); \
    hipLaunchKernelGGL(k_dbg_check_slots, dim3(1), dim3(256), 0, st, b.totals, 0u, 0u, dbg); /* counts the check */          \
  } while (0)
#else
#define PART_DBG_LEVEL(ENT, SEGS, NSEGS, REM) do { } while (0)
#endif
  const uint64_t entries = (uint64_t)p.n * p.windows;
  const dim3 scan_grid(part_ceil_div(p.nbins, 256), PART_SCAN_GROUPS);
  const uint32_t l1_grid = 8 * ((p.ntiles + 7) / 8) * p.wgroups;   // l1_tile(): a contiguous tile range per XCD; wgroups blocks per tile
  if (p.fold)
    hipLaunchKernelGGL((k_l1_hist<FR, MONT, true>), dim3(l1_grid), dim3(PART_THREADS), p.nbins * 4, st, d_scalars, d_inf, p, b.matrix);
  else
    hipLaunchKernelGGL((k_l1_hist<FR, MONT, false>), dim3(l1_grid), dim3(PART_THREADS), p.nbins * 4, st, d_scalars, d_inf, p, b.matrix);
  PART_MARK("l1_hist");
  hipLaunchKernelGGL(k_l1_scan_a, scan_grid, dim3(256), 0, st, b.matrix, p, b.partial);
  hipLaunchKernelGGL(k_l1_scan_b, dim3(1), dim3(1024), 0, st, b.partial, p, b.segs[0], b.subjob_first, b.totals);
  hipLaunchKernelGGL(k_l1_scan_c, scan_grid, dim3(256), 0, st, b.matrix, p, b.partial);
  PART_MARK("l1_scan");
  int seg_cur = 0;
  uint64_t nsegs = p.nbins;
  if (p.shared) {
    hipLaunchKernelGGL(k_l1_merge_shared, dim3(1), dim3(256), 0, st, b.segs[0], p, b.segs[1], b.subjob_first, b.totals);
    seg_cur = 1;
    nsegs = (uint64_t)p.bsets * p.b1;
  }
  if (p.fold)
    hipLaunchKernelGGL((k_l1_scatter<FR, MONT, true>), dim3(l1_grid), dim3(PART_THREADS), 0, st, d_scalars, d_inf, p, b.matrix, b.entries[0]);
  else
    hipLaunchKernelGGL((k_l1_scatter<FR, MONT, false>), dim3(l1_grid), dim3(PART_THREADS), 0, st, d_scalars, d_inf, p, b.matrix, b.entries[0]);
  PART_MARK("l1_scatter");
  PART_DBG_LEVEL(b.entries[0], b.segs[seg_cur], nsegs, p.lb);
  if (mid) (void)hipEventRecord(mid, st);
  uint32_t rb[4];
  const int np = part_pass_bits(p.lb, rb);
  uint32_t rem = p.lb;
  int cur = 0;
  for (int i = 0; i < np; i++) {
    PassPlan pp{};
    pp.nsegs = (uint32_t)nsegs;
    pp.rem = rem;
    pp.rb = rb[i];
    pp.last = (i == np - 1) ? 1 : 0;
    pp.max_subjobs = part_max_subjobs(entries, nsegs);
    PartSeg* out_segs = pp.last ? nullptr : b.segs[seg_cur ^ 1];
    // no segment can hold more entries than its window has scalars (all windows together when they share their buckets): below
    // PART_SUBJOB there are no long segments, hence no sub-jobs, and k_pass_hist / k_pass_scan have nothing to prepare
    const bool long_segs = !part_fused_takes((uint32_t)std::min<uint64_t>((uint64_t)p.n * p.levels, 0xffffffffu));
    const uint32_t gen = long_segs ? std::min<uint32_t>(pp.max_subjobs, PART_GEN_GRID) : 0;
    if (long_segs) {
      hipLaunchKernelGGL(k_pass_hist, dim3(std::min<uint32_t>(pp.max_subjobs, 2048u)), dim3(1024), 0, st, b.entries[cur], b.segs[seg_cur], b.subjob_first, pp,
                         b.counts);
      PART_MARK("pass_hist");
      hipLaunchKernelGGL(k_pass_scan, dim3(pp.nsegs), dim3(1024), 0, st, b.segs[seg_cur], b.subjob_first, pp, b.counts, out_segs);
      PART_MARK("pass_scan");
    }
    hipLaunchKernelGGL(k_pass_scatter, dim3(gen + pp.nsegs), dim3(1024), 0, st, b.entries[cur], b.segs[seg_cur], b.subjob_first, pp, gen, b.counts,
                       b.entries[cur ^ 1], out_segs);
    PART_MARK("pass_scatter");
    cur ^= 1;
    rem -= rb[i];
    if (!pp.last) {
      nsegs <<= rb[i];
      seg_cur ^= 1;
      hipLaunchKernelGGL(k_pass_subjobs, dim3(1), dim3(1024), 0, st, b.segs[seg_cur], (uint32_t)nsegs, b.subjob_first, b.totals);
      PART_DBG_LEVEL(b.entries[cur], b.segs[seg_cur], nsegs, rem);
    }
  }
#ifdef MSM_DEBUG
  {
    const uint32_t key_limit = p.bsets * p.half;
    const uint32_t idx_hi = p.idx0 + p.n + (p.levels - 1) * p.table_stride;
    if (getenv("MI355_MSM_DEBUG_CORRUPT")) hipLaunchKernelGGL(k_dbg_corrupt, dim3(1), dim3(1), 0, st, b.entries[cur], b.totals);
    hipLaunchKernelGGL(k_dbg_check_sorted, dim3(2048), dim3(256), 0, st, b.entries[cur], b.totals, key_limit, p.idx0, idx_hi, dbg);
    if (p.fold)
      hipLaunchKernelGGL((k_dbg_count_digits<FR, MONT, true>), dim3(part_ceil_div(p.n, 256)), dim3(256), 0, st, d_scalars, d_inf, p, dbg);
    else
      hipLaunchKernelGGL((k_dbg_count_digits<FR, MONT, false>), dim3(part_ceil_div(p.n, 256)), dim3(256), 0, st, d_scalars, d_inf, p, dbg);
    hipLaunchKernelGGL(k_dbg_check_slots, dim3(1), dim3(256), 0, st, b.totals, 0u, 0u, dbg);
  }
#endif
  err = hipGetLastError();
  return cur;
}

}  // namespace msm
