// msm_kernels.hpp -- the gfx950 kernels of the Pippenger pipeline.
//
//   k_convert_bases   arkworks Affine image (stride bytes, R = 2^384)  ->  device Affine (112 B, radix 2^28, R = 2^392)
//   (partition.hpp)   256-bit scalars -> signed c-bit digits -> (bucket key, base index | sign) entries grouped by key
//   k_accumulate[_glds] sorted-range walk: each lane owns K consecutive sorted entries and mixed-adds their
//                     bases; finished buckets are stored once, run fragments that cross a lane boundary
//                     go to per-lane head/tail slots
//   k_segreduce       merges the slot fragments (same walk, full XYZZ add), recursively
//   k_bucket_reduce   sum_b b * bucket[b] per window by chunked running sums, recursively
//   k_te_convert      short-Weierstrass base records -> twisted-Edwards records (BLS12-377 G1 fast path, te.hpp)
// The three walking kernels are generic over a group-law policy (laws.hpp): XYZZ for every curve, extended twisted Edwards
// for BLS12-377 G1.
//
// Reference behaviour covered: digit extraction SPK msm/pippenger.cuh:116-123, signed digits
// CMB ProcessSignedDigits.cu:118-151 / P1A mikevoronov sppark/msm/pippenger.cuh:453-479; bucket
// accumulation CMB ComputeBucketSums.cu:139-218, ML msm_kernels.cu:114-142, the sorted-range idea of
// P1A 6block cuda/mypippenger.cu:167-241; bucket reduction SPK msm/pippenger.cuh:210-244,
// CMB ReduceBuckets.cu:77-149.  The decomposition, data layout and balancing are this repo's own
// (DESIGN.md): the walk is balanced per ENTRY, not per bucket, so skewed scalar distributions
// (one hot bucket, the sparse top window) cost the same as uniform ones.
#pragma once
#include "curve.hpp"
#include "laws.hpp"
#include "msm_types.hpp"

namespace msm {

// ------------------------------------------------------------------------------------------------
// SERIALIZED = false: arkworks in-memory Affine images (Montgomery limbs, separate infinity flag byte).
// SERIALIZED = true : arkworks CanonicalSerialize uncompressed records (row f2): x | y as little-endian NORMAL-form
//                     integers, SWFlags in the top two bits of the last byte (bit 6 = infinity; P1B nickray
//                     driver/algebra/serialize/src/flags.rs:107-134, ec/.../short_weierstrass_jacobian.rs:827-835).
template <class E, bool SERIALIZED>
__global__ void __launch_bounds__(256) k_convert_bases(const uint8_t* __restrict__ in, size_t stride, uint32_t n,
                                                       AffineDevT<typename E::T>* __restrict__ out, uint8_t* __restrict__ inf) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  typename E::Md md;
  constexpr int W = E::WORDS;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(in + (size_t)i * stride);
  uint32_t w[2 * W];
#pragma unroll
  for (int k = 0; k < 2 * W; k++) w[k] = src[k];
  uint8_t flag;
  if (SERIALIZED) {
    flag = (w[2 * W - 1] >> 30) & 1;
    w[2 * W - 1] &= 0x3fffffffu;
  } else {
    flag = in[(size_t)i * stride + 8 * W];
  }
  AffineDevT<typename E::T> o;
  if (flag) {
    E::zero(o.p.x);
    E::zero(o.p.y);
  } else if (SERIALIZED) {
    E::from_plain(o.p.x, w, md);
    E::from_plain(o.p.y, w + W, md);
    E::reduce(o.p.x);
    E::reduce(o.p.y);
  } else {
    E::from_abi(o.p.x, w, md);
    E::from_abi(o.p.y, w + W, md);
    // canonical coordinates keep the "class M" contract tight and make equal points bit-identical
    E::reduce(o.p.x);
    E::reduce(o.p.y);
  }
  out[i] = o;
  inf[i] = flag ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Where a finished run fragment goes.  A fragment is the lane's partial sum for one key.  It is the
// whole bucket only if the run cannot continue into a neighbouring lane.
// `half`: which half of every Fp2 coordinate this lane holds when a point is spread over two lanes (G::LANES = 2, fp2pair.hpp);
// both lanes of such a pair walk the same entries, so everything but the stored limbs is pair-uniform.
template <class G>
__device__ __forceinline__ void seg_flush(const SegOutT<typename G::MemT>& o, uint32_t t, uint32_t nlanes, uint32_t key,
                                          const XyzzT<typename G::T>& acc, bool is_first, bool is_last, uint32_t half = 0) {
  const bool complete = (!is_first || t == 0) && (!is_last || t == nlanes - 1);
  if (complete) {
    G::store_pt(o.buckets + key, acc, half);
  } else {
    const size_t s = 2 * (size_t)t + (is_first ? 0 : 1);
    G::store_pt(o.slots + s, acc, half);
    o.slot_keys[s] = key;
  }
}

// Carried buckets WITHOUT a merge pass (SegOutT::carry_in; laws with G::CARRY_IN, i.e. the twisted-Edwards kernels).  A batch that runs
// as several chunks keeps ONE bucket array.  The run that starts bucket `key` in this chunk -- any run but a lane's first, and a lane's
// first run unless the entry before the lane has the same key (then the run started in an earlier lane, and THAT fragment took the
// stored value) -- begins from the value the earlier chunks left; the fragments of a bucket still add up to one sum (k_segreduce),
// which now includes it, and a bucket this chunk never touches keeps its value.  One 224-B read per run (~64 additions) instead of a
// pass over every bucket per chunk (k_bucket_merge: one full addition + 3 x 224 B per bucket and chunk, and a second bucket array).
// The XYZZ kernels keep the merge pass: a global load into the accumulator inside their loop costs them 65 VGPRs -- a wave per SIMD
// (165 -> 230; the two-lane G2 kernel spills) -- whatever its form (profiles/r06_ab_carry_in.txt).
template <class G>
__device__ __forceinline__ void carry_begin_run(const SegOutT<typename G::MemT>& o, uint32_t key, XyzzT<typename G::T>& acc, bool& fresh, uint32_t half) {
  // (straight into the accumulator, which is dead here: the previous run has been flushed)
  acc = G::load_pt(o.buckets + key, half);
  fresh = G::nothing(acc);        // an empty bucket: the run starts as it would without carrying
  if (fresh) G::begin_run(acc);
}

// The hot kernel.  Lane t walks sorted entries [t*K, (t+1)*K): ~K mixed adds, one bucket store per run.
// The next base is fetched before the current add so the gather latency hides under ~5k VALU ops.
// G = the group-law policy (laws.hpp); `flags[1]` is raised when the law reports a result it could not compute.
template <class G>
__global__ void __launch_bounds__(256, G::ACC_WAVES) k_accumulate(const uint2* __restrict__ entries, const uint32_t* __restrict__ n_real,
                                                    uint32_t K, const typename G::BaseDev* __restrict__ bases,
                                                    SegOutT<typename G::T> out, uint32_t nlanes, uint32_t* __restrict__ flags) {
  using E = typename G::E;
  static_assert(G::LANES == 1, "the one-lane-per-record walk has no paired form");
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nlanes) return;
  typename E::Md md;
  out.slot_keys[2 * (size_t)t] = KEY_NONE;
  out.slot_keys[2 * (size_t)t + 1] = KEY_NONE;
  // 32-bit positions: a chunk never has 2^32 entries (the engine checks), and every VGPR counts in this kernel.
  // Entries are (value = base index | sign << 31, key) pairs sorted by key (partition.hpp); their count lives on the device
  // because zero digits produce no entry.
  const uint32_t n_entries = *n_real;
  const uint64_t beg64 = (uint64_t)t * K;
  if (beg64 >= n_entries) return;
  const uint32_t beg = (uint32_t)beg64;
  const uint32_t end = (n_entries - beg > K) ? beg + K : n_entries;

  // Software pipeline: entries run TWO iterations ahead and the base gather ONE iteration ahead, so the gather
  // address never waits on an entry load (a dependent load pair would park the wave for ~1 us per add).
  uint2 ent_c = entries[beg], ent_n = make_uint2(0, KEY_NONE);
  if (end - beg > 1) ent_n = entries[beg + 1];
  // (G::PREFETCH_BASE = false -- G2, whose accumulator alone is 112 VGPRs -- gathers the base at its point of use instead)
  typename G::Base p_c;
  if (G::PREFETCH_BASE) p_c = G::from_dev(bases[ent_c.x & IDX_MASK]);

  uint32_t cur = KEY_NONE;
  bool first = true, fresh = true, bad = false;
  XyzzT<typename G::T> acc;
  G::set_identity(acc);
  for (uint32_t e = beg; e < end; e++) {
    const uint32_t key = ent_c.y, val = ent_c.x;
    if (!G::PREFETCH_BASE) p_c = G::from_dev(bases[val & IDX_MASK]);
    const typename G::Base p = p_c;
    ent_c = ent_n;
    if (G::PREFETCH_BASE && end - e > 1) p_c = G::from_dev(bases[ent_c.x & IDX_MASK]);
    if (end - e > 2) ent_n = entries[e + 2];
    if (key != cur) {
      if (cur != KEY_NONE) {
        seg_flush<G>(out, t, nlanes, cur, acc, first, false);
        first = false;
      }
      cur = key;
      fresh = true;
      G::begin_run(acc);
    }
    G::madd(acc, p, (val >> 31) != 0, fresh, md);
    if (G::CHECKS) bad |= G::failed(acc);
    fresh = false;
  }
  if (cur != KEY_NONE) seg_flush<G>(out, t, nlanes, cur, acc, first, true);
  if (G::CHECKS && bad) flags[1] = 1;
}

// The same walk with QUAD-COOPERATIVE gathers done by LDS-DMA (`global_load_lds_dwordx4`: global -> LDS, no VGPR in between).
// tools/ubench_gather.hip: when every lane of a wave loads 16 B from its own random record, the texture-address path resolves
// ~80 G lane-loads/s chip-wide -- 1.3 TB/s, 6.6 G 192-B records/s, which is where a one-lane-per-record gather of twisted-Edwards
// bases stalls (VALU 72 % busy).  When the four lanes of a quad fetch their four records TOGETHER, lane l taking bytes
// [16 l, 16 l + 16) of every 64-B sector, one wave-instruction touches 16 lines instead of 64 and the same chip gathers 24 G
// records/s (4.7 TB/s).  Round 1 staged the pieces in 32-48 VGPRs and handed them over with 8-12 ds_write_b128 per addition; G2
// could not afford the registers at all (256 VGPRs + 246 AGPRs: it gathered at the point of use and stalled ~2 us per
// addition).  With LDS-DMA (profiles/r02_ab_gather.txt: -0.7 % / -0.4 % / -3.0 % on the three curves) a wave instruction
// (record i of every quad, sector c) lands as
// ONE lane-linear kilobyte in LDS -- lane l = 4 quad + sub wrote the 16-B piece `sub` of sector c, so the 64 bytes of
// quad q's sector lie contiguous at offset 64 q -- and a lane reads its own record back sector by sector.
//   LDS per wave: 4 records x SECT sectors x 1 KB (+ a 64-B skew per record index against bank conflicts).
// Ordering: the DMA writes are covered by the wave's vmcnt (s_waitcnt vmcnt(0) before the read-back); the read-back is
// drained (lgkmcnt(0)) before the next DMA may overwrite the regions.
typedef __attribute__((address_space(3))) void* msm_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* msm_gbl_ptr_t;

template <class G>
__global__ void __launch_bounds__(256, G::ACC_WAVES) k_accumulate_glds(const uint2* __restrict__ entries, const uint32_t* __restrict__ n_real,
                                                         uint32_t K, const typename G::BaseDev* __restrict__ bases,
                                                         SegOutT<typename G::MemT> out, uint32_t nlanes, uint32_t* __restrict__ flags) {
  using E = typename G::E;
  using Base = typename G::Base;
  // LANES = 2 (SwPairLaw, fp2pair.hpp): walking lane t is the PAIR of hardware lanes 2t, 2t + 1, each holding one half of every Fp2
  // value; both run the code below on the same entries.  A quad then gathers RPQ = 2 records per iteration instead of 4.
  constexpr int LANES = G::LANES, RPQ = 4 / LANES;
  constexpr int SECT = G::GATHER_SECTORS;                  // 64-B sectors per record
  constexpr int RS = 1024;                                 // bytes of one (record index, sector) region: 64 lanes x 16 B
#ifndef MSM_ACC_LDS_PAD
#define MSM_ACC_LDS_PAD 0   // A/B only: extra LDS per wave lowers the number of resident blocks (profiles/r03_ab_occupancy.txt)
#endif
#ifndef MSM_ACC_IDX_AND
#define MSM_ACC_IDX_AND 0xffffffffu   // A/B only (profiles/r03_ab_power.txt): gather from a small subset of the records (wrong sums,
#endif                                 // same instructions) to see what the memory path costs in power, i.e. in clock
#ifndef MSM_ACC_IDX_SHL
#define MSM_ACC_IDX_SHL 0             // A/B only: spread the masked subset over the whole table (same cache footprint, full TLB footprint)
#endif
#ifndef MSM_ACC_PRIO
#define MSM_ACC_PRIO 0      // A/B only (profiles/r03_ab_setprio.txt): 1 = s_setprio 2 around the addition (a wave in its MAD-dense
#endif                      // phase runs ahead of the waves that gather), 2 = around the gather phase, 3 = odd waves start late
  constexpr int WAVE_LDS = RPQ * SECT * RS + 256 + MSM_ACC_LDS_PAD;
  static_assert(sizeof(typename G::BaseDev) % 64 == 0 && SECT >= 2 && SECT <= 4 && SECT * 64 <= (int)sizeof(typename G::BaseDev), "record layout");
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * WAVE_LDS];
  const uint32_t t = (blockIdx.x * 256 + threadIdx.x) / LANES, half = threadIdx.x % LANES;
  const uint32_t lane = threadIdx.x & 63, sub = lane & 3, quad = lane >> 2;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* wave_lds = lds + wave * WAVE_LDS;
  // my record: index `sub / LANES` within my quad; its sector c starts at rec + c * RS
  const unsigned char* rec = wave_lds + (sub / LANES) * (SECT * RS + 64) + quad * 64;
  typename E::Md md;
  const bool lane_ok = t < nlanes;
  if (lane_ok) {
    out.slot_keys[2 * (size_t)t] = KEY_NONE;
    out.slot_keys[2 * (size_t)t + 1] = KEY_NONE;
  }
  const uint32_t n_entries = *n_real;
  const uint64_t beg64 = (uint64_t)t * K;
  const bool has_work = lane_ok && beg64 < n_entries;
  const uint32_t beg = has_work ? (uint32_t)beg64 : 0;
  const uint32_t end = has_work ? ((n_entries - beg > K) ? beg + K : n_entries) : 0;

  // The sorted entries of a lane are consecutive in memory, but the lanes of a wave are K entries (2 KB) apart: an 8-byte
  // load per lane and iteration touches 64 different 64-B sectors, and by the next iteration -- 4.7 MB of base records per XCD
  // later -- the sector has left the 4-MB L2, so every entry cost a whole sector: 56 GB of the 167 GB the counters report for
  // the 2^26 launch (profiles/r03_calib_fetch.txt + r03_pmc_k_accumulate.json) against 7 GB of entries.  With EQ > 0 a lane
  // fetches 2 EQ entries at once (EQ 16-byte loads issued back to back: one request for a whole sector when EQ = 4) into a
  // register queue, refilled every 2 EQ iterations; entries are popped by a static rotation (no dynamic register indexing).
  constexpr int EQ = G::ENTRY_Q;
  uint4 q[EQ > 0 ? EQ : 1];
  uint32_t q_left = 0;   // wave-uniform
#define MSM_Q_REFILL(first_entry)                                                   \
  do {                                                                              \
    const uint4* src_ = reinterpret_cast<const uint4*>(entries + (first_entry));    \
    _Pragma("unroll") for (int j_ = 0; j_ < EQ; j_++) q[j_] = src_[j_];             \
  } while (0)
#define MSM_Q_POP(key_out, val_out)                                                 \
  do {                                                                              \
    key_out = q[0].y;                                                               \
    val_out = q[0].x;                                                               \
    _Pragma("unroll") for (int j_ = 0; j_ < EQ; j_++) {                             \
      q[j_].x = q[j_].z;                                                            \
      q[j_].y = q[j_].w;                                                            \
      if (j_ + 1 < EQ) {                                                            \
        q[j_].z = q[j_ + 1].x;                                                      \
        q[j_].w = q[j_ + 1].y;                                                      \
      }                                                                             \
    }                                                                               \
  } while (0)
  uint32_t key_c = KEY_NONE, val_c = 0, key_n = KEY_NONE, val_n = 0;
  if constexpr (EQ > 0) {
    // (entries past a lane's `end` belong to the next lane, or to the 64 bytes of slack behind the buffer; they are never used)
    if (end > beg) MSM_Q_REFILL(beg);
    MSM_Q_POP(key_c, val_c);
    MSM_Q_POP(key_n, val_n);
    q_left = 2 * EQ - 2;
    if (q_left == 0) {
      if (end - beg > 2 && end > beg) MSM_Q_REFILL(beg + 2);
      q_left = 2 * EQ;
    }
    if (end <= beg) key_c = key_n = KEY_NONE;
  } else if (end > beg) {
    const uint2 e0 = entries[beg];
    key_c = e0.y;
    val_c = e0.x;
    if (end - beg > 1) {
      const uint2 e1 = entries[beg + 1];
      key_n = e1.y;
      val_n = e1.x;
    }
  }
  bool alive = end > beg;

#define MSM_GLDS_ONE(i, ctrl)                                                                                            \
  do {                                                                                                                   \
    const uint4* s_ = reinterpret_cast<const uint4*>(bases + (uint32_t)__builtin_amdgcn_update_dpp(0, mine_, ctrl, 0xf, 0xf, true)) + sub; \
    _Pragma("unroll") for (int c_ = 0; c_ < SECT; c_++)                                                                  \
      __builtin_amdgcn_global_load_lds((msm_gbl_ptr_t)(s_ + 4 * c_), (msm_lds_ptr_t)(wave_lds + (i) * (SECT * RS + 64) + c_ * RS), 16, 0, 0); \
  } while (0)
#define MSM_GLDS_ISSUE(val, valid)                                \
  do {                                                            \
    const int mine_ = (valid) ? (int)((((val) & IDX_MASK & MSM_ACC_IDX_AND)) << MSM_ACC_IDX_SHL) : 0; \
    if constexpr (LANES == 1) {                                   \
      MSM_GLDS_ONE(0, 0x00);                                      \
      MSM_GLDS_ONE(1, 0x55);                                      \
      MSM_GLDS_ONE(2, 0xaa);                                      \
      MSM_GLDS_ONE(3, 0xff);                                      \
    } else {   /* the records of the quad's two pairs: lanes 0 and 2 name them */ \
      MSM_GLDS_ONE(0, 0x00);                                      \
      MSM_GLDS_ONE(1, 0xaa);                                      \
    }                                                             \
  } while (0)

  MSM_GLDS_ISSUE(val_c, alive);
  uint32_t cur = KEY_NONE;
  bool first = true, fresh = true, bad = false;
  XyzzT<typename G::T> acc;
  G::set_identity(acc);
#if MSM_ACC_PRIO == 3
  if (wave & 1) __builtin_amdgcn_s_sleep(127);
#endif
  uint32_t trips = K;
  if constexpr (G::ITER_BARRIER) {
    // block-uniform trip count: the block's first lane has the most entries (a lane's share is min(K, n_entries - beg))
    const uint64_t b0 = (uint64_t)blockIdx.x * (256 / LANES) * K;
    trips = b0 < n_entries ? (uint32_t)((n_entries - b0 < K) ? n_entries - b0 : K) : 0;
  }
  for (uint32_t k = 0; k < trips; k++) {
    if constexpr (G::ITER_BARRIER)
      __builtin_amdgcn_s_barrier();   // the block's waves fetch the same instructions at the same time
    else if (__builtin_amdgcn_ballot_w64(alive) == 0)
      break;   // wave-uniform: every lane of the wave has run out
    Base p;
#if MSM_ACC_PRIO == 2
    __builtin_amdgcn_s_setprio(2);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA of this entry's records has landed
    __builtin_amdgcn_wave_barrier();
    G::load_sectors(p, rec, RS, alive && (val_c >> 31) != 0, half);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // ... and has been read, before the next DMA overwrites it
    __builtin_amdgcn_wave_barrier();
    const uint32_t key = key_c, val = val_c, e = beg + k;
    const bool add_now = alive;
    key_c = key_n;
    val_c = val_n;
    alive = add_now && (end - e > 1);
#ifndef MSM_ACC_NO_GATHER   // A/B only: defined = the records gathered before the loop are reused for every addition (no base traffic)
    MSM_GLDS_ISSUE(val_c, alive);
#endif
    if constexpr (EQ > 0) {
      MSM_Q_POP(key_n, val_n);   // entry e + 2
      // the pop that empties the queue starts the refill: the loads are in flight during this iteration's addition and are
      // covered by the s_waitcnt vmcnt(0) at the top of the next one (q_left is wave-uniform: every lane pops once per iteration)
      if (--q_left == 0) {
        if (add_now && end - e > 3) MSM_Q_REFILL(e + 3);
        q_left = 2 * EQ;
      }
    } else if (add_now && end - e > 2) {
      const uint2 e2 = entries[e + 2];
      key_n = e2.y;
      val_n = e2.x;
    }
#if MSM_ACC_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
    if (add_now) {
      if (key != cur) {
        const bool lane_first = cur == KEY_NONE;
        if (cur != KEY_NONE) {
#ifndef MSM_ACC_NO_FLUSH   // A/B only (profiles/r06_ab_flush.txt): defined = a finished run is not stored (wrong sums): what the bucket stores cost
          seg_flush<G>(out, t, nlanes, cur, acc, first, false, half);
#endif
          first = false;
        }
        cur = key;
        fresh = true;
        G::begin_run(acc);
        if constexpr (G::CARRY_IN) {
          // (rare: once per run, and only in the later chunks of a carried batch; the entry before the lane is read here, not kept in a register)
          if (out.carry_in && !(lane_first && t > 0 && entries[beg - 1].y == key)) carry_begin_run<G>(out, key, acc, fresh, half);
        }
      }
#if MSM_ACC_PRIO == 1
      __builtin_amdgcn_s_setprio(2);
#endif
      if (G::madd_loaded(acc, p, (val >> 31) != 0, fresh, md)) {
        // exceptional pair (short Weierstrass only): the record in LDS is already being overwritten, so fetch it again
        const Base again = G::from_dev_lane(bases[val & IDX_MASK], half);
        G::madd_same_x(acc, again, (val >> 31) != 0, md);
      }
#if MSM_ACC_PRIO == 1
      __builtin_amdgcn_s_setprio(0);
#endif
      if (G::CHECKS) bad |= G::failed(acc);
      fresh = false;
    }
  }
  if (cur != KEY_NONE) seg_flush<G>(out, t, nlanes, cur, acc, first, true, half);
  if (G::CHECKS && bad) flags[1] = 1;
#undef MSM_GLDS_ISSUE
#undef MSM_GLDS_ONE
#undef MSM_Q_REFILL
#undef MSM_Q_POP
}

// The plain sum of bases [first, first + n) that are not flagged infinite -- what an anchored window's constant part is a multiple of
// (msm_engine.hip, "the anchored window").  Lane t adds its `per_lane` consecutive bases with the law's mixed addition and leaves ONE
// fragment of key 0 in slot 2t; the fragment merge (k_segreduce) adds the fragments up into out.buckets[0].  Runs once per context
// and number of pairs: no gather tricks, the next record is simply loaded before the current addition.
template <class G>
__global__ void __launch_bounds__(256, G::ACC_WAVES) k_sum_bases(const typename G::BaseDev* __restrict__ bases, const uint8_t* __restrict__ inf,
                                                   uint32_t first, uint32_t n, uint32_t per_lane, SegOutT<typename G::T> out, uint32_t nlanes,
                                                   uint32_t* __restrict__ flags) {
  using E = typename G::E;
  static_assert(G::LANES == 1, "one lane per point");
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nlanes) return;
  typename E::Md md;
  const uint64_t beg64 = (uint64_t)t * per_lane;
  const uint32_t beg = beg64 < n ? (uint32_t)beg64 : n, end = (n - beg > per_lane) ? beg + per_lane : n;
  bool fresh = true, bad = false;
  XyzzT<typename G::T> acc;
  G::set_identity(acc);
  G::begin_run(acc);
  typename G::Base p_n;
  if (beg < end) p_n = G::from_dev(bases[first + beg]);
  for (uint32_t i = beg; i < end; i++) {
    const typename G::Base p = p_n;
    const bool dead = inf[first + i] != 0;
    if (i + 1 < end) p_n = G::from_dev(bases[first + i + 1]);
    if (dead) continue;
    G::madd(acc, p, false, fresh, md);
    if (G::CHECKS) bad |= G::failed(acc);
    fresh = false;
  }
  out.slot_keys[2 * (size_t)t + 1] = KEY_NONE;
  if (fresh) {
    out.slot_keys[2 * (size_t)t] = KEY_NONE;
  } else {
    G::store_pt(out.slots + 2 * (size_t)t, acc, 0);
    out.slot_keys[2 * (size_t)t] = 0;
  }
  if (G::CHECKS && bad) flags[1] = 1;
}

// Merge run fragments: same walk over the slot sequence of the previous level (keys non-decreasing,
// KEY_NONE = hole), full additions.  Recursion ends when one lane covers everything.
template <class G>
__global__ void __launch_bounds__(256) k_segreduce(const XyzzDevT<typename G::MemT>* __restrict__ in_slots,
                                                   const uint32_t* __restrict__ in_keys, uint32_t n_in, uint32_t K,
                                                   SegOutT<typename G::MemT> out, uint32_t nlanes, uint32_t* __restrict__ flags) {
  using E = typename G::E;
  const uint32_t t = (blockIdx.x * 256 + threadIdx.x) / G::LANES, half = threadIdx.x % G::LANES;
  if (t >= nlanes) return;
  typename E::Md md;
  out.slot_keys[2 * (size_t)t] = KEY_NONE;
  out.slot_keys[2 * (size_t)t + 1] = KEY_NONE;
  const uint64_t beg = (uint64_t)t * K;
  const uint64_t end = (beg + K < n_in) ? beg + K : n_in;
  uint32_t cur = KEY_NONE;
  bool first = true, bad = false;
  XyzzT<typename G::T> acc;
  G::set_identity(acc);
  for (uint64_t e = beg; e < end; e++) {
    const uint32_t key = in_keys[e];
    if (key == KEY_NONE) continue;
    const XyzzT<typename G::T> v = G::load_pt(in_slots + e, half);
    if (key != cur) {
      if (cur != KEY_NONE) {
        seg_flush<G>(out, t, nlanes, cur, acc, first, false, half);
        first = false;
      }
      cur = key;
      acc = v;
    } else {
      G::add(acc, v, md);
      if (G::CHECKS) bad |= G::failed(acc);
    }
  }
  if (cur != KEY_NONE) seg_flush<G>(out, t, nlanes, cur, acc, first, true, half);
  if (G::CHECKS && bad) flags[1] = 1;
}

// ------------------------------------------------------------------------------------------------
// Bucket -> window reduction, one level.  Per window the target is
//     V = sum_j A_j + sum_j weight(j) * X_j,    weight(j) = j + 1 on the first level, j afterwards,
// with X = buckets and no A on the first level.  Thread (w, t) owns chunk j in [tL, tL+L): it emits
//     A'_t = sum A_j + sum (local weight) X_j     and     X'_t = L * sum X_j
// so that V = sum_t A'_t + sum_t t * X'_t -- the same problem, L times smaller.  When one chunk is
// left, V = A'_0.  Running sums walk the chunk from the top: run += X_j; wsum += run.
// L need not be a power of two: the first level of a large window is cut so that its chunks fill the chip's SIMDs with exactly one
// wave each (Plan::L0) -- 13 windows x 4096 chunks of 128 buckets are 832 waves on 1024 SIMDs, 4994 chunks of 105 are 1015; the
// outputs of window w start at w * out_stride (a power of two for the scan that follows, the tail of a row stays empty).
template <class G, bool FIRST>
__global__ void __launch_bounds__(256, G::ACC_WAVES) k_bucket_reduce(const XyzzDevT<typename G::MemT>* __restrict__ in_a,
                                                       const XyzzDevT<typename G::MemT>* __restrict__ in_x,
                                                       uint32_t n_per_win, uint32_t L, uint32_t chunks_per_win,
                                                       uint32_t windows, uint32_t out_stride, XyzzDevT<typename G::MemT>* __restrict__ out_a,
                                                       XyzzDevT<typename G::MemT>* __restrict__ out_x, uint32_t* __restrict__ flags) {
  using E = typename G::E;
  using XD = XyzzDevT<typename G::MemT>;
  const uint32_t g = (blockIdx.x * 256 + threadIdx.x) / G::LANES, half = threadIdx.x % G::LANES;
  if (g >= windows * chunks_per_win) return;
  typename E::Md md;
  const uint32_t w = g / chunks_per_win, t = g % chunks_per_win;
  const uint32_t lo = t * L;
  const uint32_t hi = (lo + L < n_per_win) ? lo + L : n_per_win;
  const XD* x = in_x + (size_t)w * n_per_win;
  XyzzT<typename G::T> run, wsum;
  bool bad = false;
  G::set_identity(run);
  G::set_identity(wsum);
#ifndef MSM_REDUCE_PREFETCH
#define MSM_REDUCE_PREFETCH 0   // A/B (profiles/r05_ab_reduce_prefetch.txt): 1 = the next bucket is loaded before the two additions of the current one
#endif
#if MSM_REDUCE_PREFETCH
  XyzzT<typename G::T> v_next;
  if (hi > lo) v_next = G::load_pt(x + hi - 1, half);
  for (uint32_t j = hi; j-- > lo;) {
    const XyzzT<typename G::T> v = v_next;
    if (j > lo) v_next = G::load_pt(x + j - 1, half);
    G::add(run, v, md);
    if (G::CHECKS) bad |= G::failed(run);
    if (FIRST || j > lo) {
      G::add(wsum, run, md);
      if (G::CHECKS) bad |= G::failed(wsum);
    }
  }
#else
  for (uint32_t j = hi; j-- > lo;) {
    const XyzzT<typename G::T> v = G::load_pt(x + j, half);
    G::add(run, v, md);
    if (G::CHECKS) bad |= G::failed(run);
    if (FIRST || j > lo) {
      G::add(wsum, run, md);
      if (G::CHECKS) bad |= G::failed(wsum);
    }
  }
#endif
  if (!FIRST) {
    const XD* a = in_a + (size_t)w * n_per_win;
    for (uint32_t j = lo; j < hi; j++) {
      const XyzzT<typename G::T> v = G::load_pt(a + j, half);
      G::add(wsum, v, md);
      if (G::CHECKS) bad |= G::failed(wsum);
    }
  }
  const size_t at = (size_t)w * out_stride + t;
  G::store_pt(out_a + at, wsum, half);
  // X'_t = L * run, most significant bit of L first: doublings, and an addition of the sum itself for every further set bit
  if ((L & (L - 1)) == 0) {
    G::mul_pow2(run, 31 - __builtin_clz(L), md);
    if (G::CHECKS) bad |= G::failed(run);
  } else {
    const XyzzT<typename G::T> one = run;
    for (int b = 30 - __builtin_clz(L); b >= 0; b--) {
      G::mul_pow2(run, 1, md);
      if (G::CHECKS) bad |= G::failed(run);
      if ((L >> b) & 1) {
        G::add(run, one, md);
        if (G::CHECKS) bad |= G::failed(run);
      }
    }
  }
  G::store_pt(out_x + at, run, half);
  if (G::CHECKS && bad) flags[1] = 1;
}

// ------------------------------------------------------------------------------------------------
// Carried buckets: a batch that runs as several chunks (the first piece of a host-scalar batch that is computed while the rest
// crosses PCIe, the slices of the stateless pipeline, chunks sized to the free memory) keeps ONE bucket array for the batch and
// reduces it once: total[b] += part[b] after every chunk but the first.  One full addition per bucket and chunk (6.8 M at c = 20)
// instead of a bucket reduction (two per bucket, then the scan tail), a host fold and a stream synchronisation per chunk -- and
// every chunk can use the window size of the WHOLE batch.
template <class G>
__global__ void __launch_bounds__(256, G::ACC_WAVES) k_bucket_merge(XyzzDevT<typename G::MemT>* __restrict__ total,
                                                                    const XyzzDevT<typename G::MemT>* __restrict__ part, uint32_t n,
                                                                    uint32_t* __restrict__ flags) {
  using E = typename G::E;
  const uint32_t g = (blockIdx.x * 256 + threadIdx.x) / G::LANES, half = threadIdx.x % G::LANES;
  if (g >= n) return;
  const XyzzT<typename G::T> v = G::load_pt(part + g, half);
  if (G::nothing(v)) return;   // the chunk left this bucket empty
  XyzzT<typename G::T> t = G::load_pt(total + g, half);
  if (G::nothing(t)) {
    G::store_pt(total + g, v, half);
    return;
  }
  typename E::Md md;
  G::add(t, v, md);
  if (G::CHECKS && G::failed(t)) flags[1] = 1;
  G::store_pt(total + g, t, half);
}

// ------------------------------------------------------------------------------------------------
// Bucket -> window reduction for SMALL windows (<= 4096 buckets), as a parallel scan: the chunked running sums above are
// work-efficient but ~80 full additions deep at 2^16 pairs, and a lone wave needs ~10 us per addition -- the bucket
// reduction was 40 % of a small MSM.  Here every step is ONE addition per thread:
//   suffix scan   S_j <- S_j + S_(j+d),  d = 1, 2, 4, ...        after log2(nb) steps  S_j = sum_(b >= j) B_b
//   tree          S_j <- S_j + S_(j+h),  h = nb/2, nb/4, ..., 1   after log2(nb) steps  S_0 = sum_j S_j = sum_b (b+1) B_b
// 2 log2(nb) launches (22 at 2^16) of n log n work in total, which is nothing at these sizes.
// mode 0: scan step (all j, partner j + d if it exists); mode 1: tree step (j < d, partner j + d).
// Larger windows first run ONE level of the chunked scheme (k_bucket_reduce<FIRST>: chunk t of L buckets -> A_t, X_t with
// V = sum_t A_t + sum_t t X_t), which leaves <= 4096 chunks per window; the scan then runs on the X_t, and
// mode 2 joins the two sums: out_j = (j == 0 ? identity : S_j) + A_j   (sum_t t X_t = sum_(j >= 1) S_j), a tree finishes.
template <class G>
__global__ void __launch_bounds__(256, G::ACC_WAVES) k_reduce_scan_step(const XyzzDevT<typename G::MemT>* __restrict__ in,
                                                                        const XyzzDevT<typename G::MemT>* __restrict__ in2,
                                                                        XyzzDevT<typename G::MemT>* __restrict__ out, uint32_t nb, uint32_t windows,
                                                                        uint32_t d, uint32_t mode, uint32_t* __restrict__ flags) {
  using E = typename G::E;
  using XD = XyzzDevT<typename G::MemT>;
  const uint32_t span = mode == 1 ? d : nb;
  const uint32_t g = (blockIdx.x * 256 + threadIdx.x) / G::LANES, half = threadIdx.x % G::LANES;
  if (g >= windows * span) return;
  typename E::Md md;
  const uint32_t w = g / span, j = g % span;
  const XD* row = in + (size_t)w * nb;
  XyzzT<typename G::T> r = G::load_pt(row + j, half);
  if (G::is_empty(r) || (mode == 2 && j == 0)) G::set_identity(r);     // (a bucket nobody wrote is all zero)
  bool bad = false;
  if (mode == 2) {
    const XyzzT<typename G::T> v = G::load_pt(in2 + (size_t)w * nb + j, half);
    G::add(r, v, md);
    if (G::CHECKS) bad = G::failed(r);
  } else if (j + d < nb) {
    const XyzzT<typename G::T> v = G::load_pt(row + j + d, half);
    G::add(r, v, md);
    if (G::CHECKS) bad = G::failed(r);
  }
  G::store_pt(out + (size_t)w * nb + j, r, half);
  if (G::CHECKS && bad) flags[1] = 1;
}

// ------------------------------------------------------------------------------------------------
// The two latency-bound kernels again, FOUR LANES PER ADDITION (te.hpp te_add_quad, curve.hpp xyzz_add_quad): a step of
// the scan reduction or a level of the fragment merge on a small input is one dependent addition per wave (~14 us for a lone
// wave at one instruction per ~5.5 cycles) on a chip that is otherwise idle; with the nine multiplications spread over a quad
// it is three (Edwards) or four (XYZZ) multiplications deep.  Lane q of a quad owns coordinate q (the order of XyzzT in memory) of the
// operands and of the result.  Same semantics as k_reduce_scan_step / k_segreduce, which stay in charge above
// LaunchTe::quad_limit additions per launch, where throughput counts (the quad form spends 4/3 of the instructions).
// (The first chunked level of the bucket reduction -- 53 K chunks of 128 buckets at 2^26 pairs -- looks latency-bound too but
// is not: a quad per chunk made it 3.0 -> 3.9 ms.)
template <class T>
__device__ __forceinline__ const T& xyzz_coord(const XyzzDevT<T>* pts, size_t i, uint32_t q) { return reinterpret_cast<const T*>(pts + i)[q]; }
template <class T>
__device__ __forceinline__ T& xyzz_coord(XyzzDevT<T>* pts, size_t i, uint32_t q) { return reinterpret_cast<T*>(pts + i)[q]; }

// What the quad kernels need of a group law: the identity's coordinate q, "this slot was never written" (quad-uniform), and
// a += b on coordinates (returns true when the law reports a result it could not compute).
template <class F>
struct TeQuad {   // extended twisted Edwards (te.hpp te_add_quad): identity (0, 1, 1, 0); Z = 0 marks an empty bucket / a failure
  using T = Fe;
  using Md = Modulus<F>;
  static __device__ __forceinline__ void identity(Fe& r, uint32_t q) {
    Fe one, zero;
    fe_set(one, F::ONE);
    fe_zero(zero);
    fe_select(r, zero, one, q == 1 || q == 2);
  }
  static __device__ __forceinline__ bool is_empty(const Fe& r) { return quad_flag<0xAA>(fe_is_zero_M<F>(r)); }
  static __device__ __forceinline__ bool add(Fe& a, const Fe& b, uint32_t q, const Modulus<F>& md) {
    if (is_empty(b)) return false;
    te_add_quad<F>(a, b, q, md);
    return q == 2 && fe_is_zero_M<F>(a);
  }
};
template <class E>
struct SwQuad {   // XYZZ (curve.hpp xyzz_add_quad), Fp or Fp2 coordinates: the all-zero point IS the identity, the law has no failures
  using T = typename E::T;
  using Md = typename E::Md;
  static __device__ __forceinline__ void identity(T& r, uint32_t) { E::zero(r); }
  static __device__ __forceinline__ bool is_empty(const T&) { return false; }
  static __device__ __forceinline__ bool add(T& a, const T& b, uint32_t q, const Md& md) {
    xyzz_add_quad<E>(a, b, q, md);
    return false;
  }
};

template <class Q>
__global__ void __launch_bounds__(256) k_reduce_scan_step_quad(const XyzzDevT<typename Q::T>* __restrict__ in, const XyzzDevT<typename Q::T>* __restrict__ in2,
                                                               XyzzDevT<typename Q::T>* __restrict__ out,
                                                               uint32_t nb, uint32_t windows, uint32_t d, uint32_t mode, uint32_t* __restrict__ flags) {
  using T = typename Q::T;
  const uint32_t span = mode == 1 ? d : nb;
  const uint32_t gq = blockIdx.x * 256 + threadIdx.x, g = gq >> 2, q = gq & 3;
  if (g >= windows * span) return;
  typename Q::Md md;
  const uint32_t w = g / span, j = g % span;
  const size_t row = (size_t)w * nb;
  T r = xyzz_coord(in, row + j, q);
  if (Q::is_empty(r) || (mode == 2 && j == 0)) Q::identity(r, q);   // (a bucket nobody wrote is all zero)
  const bool have = mode == 2 || j + d < nb;
  if (have) {
    const T v = mode == 2 ? xyzz_coord(in2, row + j, q) : xyzz_coord(in, row + j + d, q);
    if (Q::add(r, v, q, md)) flags[1] = 1;
  }
  xyzz_coord(out, row + j, q) = r;
}

template <class T>
__device__ __forceinline__ void seg_flush_quad(const SegOutT<T>& o, uint32_t t, uint32_t nlanes, uint32_t key, const T& acc, uint32_t q, bool is_first,
                                               bool is_last) {
  const bool complete = (!is_first || t == 0) && (!is_last || t == nlanes - 1);
  if (complete) {
    xyzz_coord(o.buckets, key, q) = acc;
  } else {
    const size_t s = 2 * (size_t)t + (is_first ? 0 : 1);
    xyzz_coord(o.slots, s, q) = acc;
    if (q == 0) o.slot_keys[s] = key;
  }
}

template <class Q>
__global__ void __launch_bounds__(256) k_segreduce_quad(const XyzzDevT<typename Q::T>* __restrict__ in_slots, const uint32_t* __restrict__ in_keys, uint32_t n_in, uint32_t K,
                                                        SegOutT<typename Q::T> out, uint32_t nlanes, uint32_t* __restrict__ flags) {
  using T = typename Q::T;
  const uint32_t gq = blockIdx.x * 256 + threadIdx.x, t = gq >> 2, q = gq & 3;
  if (t >= nlanes) return;
  typename Q::Md md;
  if (q == 0) {
    out.slot_keys[2 * (size_t)t] = KEY_NONE;
    out.slot_keys[2 * (size_t)t + 1] = KEY_NONE;
  }
  const uint64_t beg = (uint64_t)t * K;
  const uint64_t end = (beg + K < n_in) ? beg + K : n_in;
  uint32_t cur = KEY_NONE;
  bool first = true, bad = false;
  T acc;
  Q::identity(acc, q);
  for (uint64_t e = beg; e < end; e++) {
    const uint32_t key = in_keys[e];
    if (key == KEY_NONE) continue;
    const T v = xyzz_coord(in_slots, e, q);
    if (key != cur) {
      if (cur != KEY_NONE) {
        seg_flush_quad(out, t, nlanes, cur, acc, q, first, false);
        first = false;
      }
      cur = key;
      acc = v;
    } else {
      bad |= Q::add(acc, v, q, md);
    }
  }
  if (cur != KEY_NONE) seg_flush_quad(out, t, nlanes, cur, acc, q, first, true);
  if (bad) flags[1] = 1;
}

// ------------------------------------------------------------------------------------------------
// Fixed-base precompute (row f1): table level w holds 2^(c w) * P_i as affine points, so every digit of a scalar feeds
// ONE shared bucket set and the bucket->window reduction and the Horner fold shrink by the number of windows
// (CMB PrecomputePoints.cu:10-39 builds 2^(46k) P the same way; P1A matter-labs/src/lib.rs:101-114).
// Step 1: XYZZ = 2^c * (level w-1), one thread per point.
template <class E>
__global__ void __launch_bounds__(256) k_pre_double(const AffineDevT<typename E::T>* __restrict__ in, const uint8_t* __restrict__ inf_in,
                                                    uint32_t n, uint32_t c, XyzzDevT<typename E::T>* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  typename E::Md md;
  XyzzDevT<typename E::T> o;
  if (inf_in[i]) {
    xyzz_set_inf<E>(o.p);
  } else {
    const AffineDevT<typename E::T> a = in[i];
    xyzz_from_affine<E>(o.p, a.p, false);
    for (uint32_t k = 0; k < c; k++) xyzz_dbl<E>(o.p, md);
  }
  out[i] = o;
}

// Step 2: back to affine with one inversion per lane (Montgomery's trick over J consecutive points): forward pass stores
// the running products of zz*zzz, the backward pass peels one inverse per point.  A point that became infinity
// (a low-order base doubled away) is written as zeros and flagged.
template <class E>
__global__ void __launch_bounds__(256) k_pre_normalize(const XyzzDevT<typename E::T>* __restrict__ in, uint32_t n, uint32_t J,
                                                       typename E::T* __restrict__ prefix, AffineDevT<typename E::T>* __restrict__ out,
                                                       uint8_t* __restrict__ inf_out) {
  using El = typename E::T;
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint64_t lo = (uint64_t)t * J;
  if (lo >= n) return;
  const uint64_t hi = (lo + J < n) ? lo + J : n;
  typename E::Md md;
  El run;
  E::set_one(run);
  for (uint64_t j = lo; j < hi; j++) {
    const XyzzDevT<El> v = in[j];
    prefix[j] = run;
    if (!xyzz_is_inf<E>(v.p)) {
      El z;
      E::mul(z, v.p.zz, v.p.zzz, md);
      E::mul(run, run, z, md);
    }
  }
  El inv;
  el_inv(inv, run, md, (E*)nullptr);
  for (uint64_t j = hi; j-- > lo;) {
    const XyzzDevT<El> v = in[j];
    AffineDevT<El> o;
    if (xyzz_is_inf<E>(v.p)) {
      E::zero(o.p.x);
      E::zero(o.p.y);
      inf_out[j] = 1;
    } else {
      El z, ti, zzi, zzzi;
      const El pre = prefix[j];
      E::mul(ti, inv, pre, md);          // (zz_j zzz_j)^-1
      E::mul(z, v.p.zz, v.p.zzz, md);
      E::mul(inv, inv, z, md);
      E::mul(zzi, ti, v.p.zzz, md);
      E::mul(zzzi, ti, v.p.zz, md);
      E::mul(o.p.x, v.p.x, zzi, md);
      E::mul(o.p.y, v.p.y, zzzi, md);
      E::reduce(o.p.x);
      E::reduce(o.p.y);
      inf_out[j] = 0;
    }
    out[j] = o;
  }
}

// ------------------------------------------------------------------------------------------------
// Short-Weierstrass device records -> twisted-Edwards records (te.hpp), Montgomery's trick over J consecutive points per
// lane: the forward pass stores the running products of the map's denominators, the backward pass peels one inverse per
// point.  Points without an image are counted in flags[0] (the engine then keeps the base set on the XYZZ path) and get a
// harmless filler; bases flagged infinite are never gathered.
// The map runs in the 14 x 28 shape of F (init-time work); the records are written in the limb shape of TF, the field the
// Edwards kernels compute in (te_record_to).
template <class F, class TF>
__global__ void __launch_bounds__(256) k_te_convert(const AffineDev* __restrict__ in, const uint8_t* __restrict__ inf, uint32_t n, uint32_t J,
                                                    Fe* __restrict__ prefix, TeAffineDev* __restrict__ out, uint32_t* __restrict__ flags) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint64_t lo = (uint64_t)t * J;
  if (lo >= n) return;
  const uint64_t hi = (lo + J < n) ? lo + J : n;
  Modulus<F> md;
  Modulus<TF> tmd;
  Fe run;
  fe_set(run, F::ONE);
  uint32_t bad = 0;
  for (uint64_t j = lo; j < hi; j++) {
    prefix[j] = run;
    if (inf[j]) continue;
    const AffineDev a = in[j];
    Fe u, v, w, den;
    te_map_prepare<F>(u, v, w, den, a.p, md);
    if (fe_is_zero_M<F>(den))
      bad++;
    else
      fe_mul<F>(run, run, den, md);
  }
  Fe inv;
  fe_inv<F>(inv, run, md);
  for (uint64_t j = hi; j-- > lo;) {
    TeAffine o;
    fe_set(o.ymx, TF::ONE);   // the identity (0, 1): a harmless filler
    fe_set(o.ypx, TF::ONE);
    fe_zero(o.td);
    if (!inf[j]) {
      const AffineDev a = in[j];
      Fe u, v, w, den;
      te_map_prepare<F>(u, v, w, den, a.p, md);
      if (!fe_is_zero_M<F>(den)) {
        Fe ti;
        const Fe pre = prefix[j];
        fe_mul<F>(ti, inv, pre, md);      // 1 / den_j
        fe_mul<F>(inv, inv, den, md);
        TeAffine o28;
        te_map_finish<F>(o28, u, v, w, ti, md);
        te_record_to<TF>(o, o28, tmd);
      }
    }
    TeAffineDev od;
    od.set(o);
    out[j] = od;
  }
  if (bad) atomicAdd(&flags[0], bad);
}

}  // namespace msm
