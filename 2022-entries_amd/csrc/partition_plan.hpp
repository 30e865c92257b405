// partition_plan.hpp -- geometry of the bucket grouping (partition.hpp): plain structs and host arithmetic, no kernels, so the
// engine can size buffers and plan launches without instantiating device code.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <algorithm>

namespace msm {

constexpr int PART_TILE = 8192;        // scalars per level-1 tile (8 per thread of a 1024-thread block)
constexpr int PART_THREADS = 1024;
constexpr int PART_PER_THREAD = PART_TILE / PART_THREADS;
// 10 bits at level 1 (from 2^25 scalars on, see by_size below) leave c = 21 with ONE generic pass instead of two (sort 12.0 -> 6.8 ms
// at 2^26) and c = 20 with a 9-bit one (sort 6.7 -> 5.8, level 1 4.4 -> 4.9: -0.3 ms); tools/ab_fold.sh, profiles/r03_ab_fold.txt
#ifndef PART_MAX_HB_V
#define PART_MAX_HB_V 10
#endif
constexpr int PART_MAX_HB = PART_MAX_HB_V;   // level-1 bins per window: 2^HB <= 1024 = the block size of the level-1 kernels
constexpr int PART_MAX_RB = 10;        // bins of one generic pass: 2^RB <= 1024
constexpr uint32_t PART_SUBJOB = 192u << 10;   // entries per sub-job of a generic pass
#ifndef PART_PTILE_V
#define PART_PTILE_V 16384
#endif
#ifndef PART_PASS_OCC
#define PART_PASS_OCC 4
#endif
// 16384-entry tiles (128 KB of LDS, one block per CU) give 16-entry / 128-B runs at 1024 bins; measured at 2^26 against 8192-entry
// tiles with two blocks per CU: 6.5 vs 7.5 ms for the pass (profiles/r02_partition_tiles.txt)
constexpr int PART_PTILE = PART_PTILE_V;       // entries per LDS-staged tile of a generic pass
constexpr int PART_SCAN_GROUPS = 64;   // tile groups of the level-1 column scan

struct PartSeg {          // a run of entries that share their high key bits
  uint32_t start, len;    // position in the entry array
  uint32_t key_base;      // key bits already resolved (full key = key_base + the bits still in the entries' high word)
  uint32_t pad;
};

// Geometry of one grouping problem, filled by the host (part_plan in msm_engine.hip).
struct PartPlan {
  uint32_t n, c, windows, half;      // scalars, window bits, digit windows, 2^(c-1)
  uint32_t shared;                   // 1: precomputed tables -- windows that differ by a multiple of `bsets` share a bucket set
  uint32_t levels, bsets;            // table levels k and bucket sets G = ceil(windows / k): window w = g + G j reads table level j (base index
                                     // + j * table_stride) into bucket set g.  No tables: levels = 1, bsets = windows.  Tables for every window
                                     // (the round-1..3 form): levels = windows, bsets = 1.
  uint32_t idx0, table_stride;       // base index of scalar 0, distance between table levels
  uint32_t hb, lb;                   // bucket bits resolved by level 1 / left after it (hb + lb = c - 1)
  uint32_t b1;                       // 2^hb
  uint32_t nbins;                    // level-1 bins = segments after level 1: windows * b1
  uint32_t ntiles;                   // ceil(n / PART_TILE)
  uint32_t tiles_per_group;          // column-scan grouping
  uint32_t wgroups, wper;            // level-1 blocks per tile (each takes `wper` consecutive windows): fills the chip when tiles are few
  uint32_t fold;                     // 1: a scalar k in (r/2, r) is replaced by r - k with all its digits negated (load_scalar)
  uint32_t anchor;                   // the window whose digit is taken relative to 2^(c-1) and ends the carry chain (next_digit); PART_NO_ANCHOR: none
};
constexpr uint32_t PART_NO_ANCHOR = 0xffffffffu;

// Geometry of one generic pass.
struct PassPlan {
  uint32_t nsegs;       // segments coming in
  uint32_t rem, rb;     // key bits still unresolved in the entries / resolved by this pass
  uint32_t last;        // 1: write the full key into the high word (rem == rb)
  uint32_t max_subjobs; // grid size (upper bound; the real count is in totals[1])
};


inline uint32_t part_ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// `levels` = precomputed table levels (0 or 1: none).  With k levels the windows g, g + G, g + 2G, ... (G = ceil(windows / k)) feed bucket
// set g: yrrid's shape is k = 6, G = 2 (CMB PrecomputePoints.cu:10-39, MSM.cu:380-383); k >= windows is "one bucket set for all".
inline PartPlan part_plan(uint32_t n, uint32_t c, uint32_t windows, uint32_t levels, uint32_t idx0, uint32_t table_stride, bool fold = false,
                          uint32_t anchor = 0xffffffffu) {
  PartPlan p{};
  p.fold = fold ? 1 : 0;
  p.anchor = fold ? 0xffffffffu : anchor;
  p.n = n;
  p.c = c;
  p.windows = windows;
  p.half = 1u << (c - 1);
  p.levels = levels > 1 ? std::min(levels, windows) : 1;
  p.bsets = part_ceil_div(windows, p.levels);
  p.levels = part_ceil_div(windows, p.bsets);      // (no level without a window: 13 windows in 6 levels are 3 sets x 5 levels)
  p.shared = p.levels > 1 ? 1 : 0;
  p.idx0 = idx0;
  p.table_stride = table_stride;
  // Level 1 resolves as many bucket bits as it can (it is fused with the digit extraction, so its bits are the cheap ones)
  // without cutting the input into segments of less than ~2^15 entries: a small MSM would otherwise pay for thousands of
  // near-empty blocks in the passes that follow (2^16 pairs: 0.65 ms of grouping with 512 bins per window, 0.1 ms with 2).
  const uint32_t bits = c - 1;
  uint32_t lg = 0;
  while ((2ull << lg) <= (uint64_t)n) lg++;                 // floor(log2 n) for n >= 1
  const uint32_t by_size = lg > 15 ? lg - 15 : 0;
  p.hb = std::min<uint32_t>(std::min<uint32_t>(bits, (uint32_t)PART_MAX_HB), by_size);
  // ... but the bits it leaves must fit the low half of an entry's key word (the bin rides in the high half while an entry is
  // staged in LDS): a small input with a large forced window would otherwise leave up to 23
  if (bits - p.hb > 15) p.hb = bits - 15;
  // ... and a level-1 bit is cheaper than a whole extra generic pass (every pass moves all entries once more): a chunk of a carried
  // batch is small but runs at the window size of the batch -- 5 M pairs at c = 20 would leave 12 bits, two passes
  if (bits - p.hb > (uint32_t)PART_MAX_RB && bits - (uint32_t)PART_MAX_RB <= (uint32_t)PART_MAX_HB && bits - p.hb <= 2 * (uint32_t)PART_MAX_RB)
    p.hb = bits - (uint32_t)PART_MAX_RB;
  p.lb = bits - p.hb;
  p.b1 = 1u << p.hb;
  p.nbins = p.bsets * p.levels * p.b1;   // (= windows * b1 without tables; with them a few bins of the last level may stay empty)
  p.ntiles = part_ceil_div(n, PART_TILE);
  if (p.ntiles == 0) p.ntiles = 1;
  p.tiles_per_group = part_ceil_div(p.ntiles, PART_SCAN_GROUPS);
  // A level-1 block walks its windows one after the other (~5 us each): with few tiles, give every tile several blocks that
  // split the windows between them (the scalars are read once more per block; at these sizes that is nothing)
  const uint32_t want = p.ntiles >= 256 ? 1 : part_ceil_div(256, p.ntiles);
  p.wgroups = std::min<uint32_t>(windows, want);
  p.wper = part_ceil_div(windows, p.wgroups);
  p.wgroups = part_ceil_div(windows, p.wper);
  return p;
}

// Bits taken by each generic pass after level 1: as few passes as possible, at most PART_MAX_RB bits each.  The LAST pass takes
// as many bits as it may and the earlier ones share the rest: a pass costs per SEGMENT it is given (a block per sub-job), so
// the pass that fans out widest must come last -- 6 + 6 bits at c = 22 left 426 K segments of 2 K entries to the second pass
// (19 ms); 2 + 10 leaves it 26 K segments of 32 K entries.
// Always at least one pass (it is the pass that writes the full key into the entries), possibly over zero bits.
inline int part_pass_bits(uint32_t lb, uint32_t (&rb)[4]) {
  int np = (int)((lb + PART_MAX_RB - 1) / PART_MAX_RB);
  if (np == 0) np = 1;
  uint32_t last = lb < (uint32_t)PART_MAX_RB ? lb : (uint32_t)PART_MAX_RB;
  uint32_t left = lb - last;
  for (int i = 0; i < np - 1; i++) {
    rb[i] = (left + (uint32_t)(np - 1 - i) - 1) / (uint32_t)(np - 1 - i);
    left -= rb[i];
  }
  rb[np - 1] = last;
  return np;
}

// A segment of at most PART_SUBJOB entries is ONE job with nothing to coordinate between blocks: the fused kernel (k_pass_fused)
// takes it whole -- every segment of a uniform input -- and it counts NO sub-jobs; only longer segments (a skewed input: all
// scalars equal puts a whole window into one segment) are cut into sub-jobs for the three generic kernels.
#ifdef PART_NO_FUSED_PASS
__host__ __device__ inline bool part_fused_takes(uint32_t) { return false; }
__host__ __device__ inline uint32_t part_subjobs_of(uint32_t len) { return (len + PART_SUBJOB - 1) / PART_SUBJOB; }
// Upper bound of the sub-jobs of a pass over `nsegs` segments holding `entries` entries in total.
inline uint32_t part_max_subjobs(uint64_t entries, uint64_t nsegs) { return (uint32_t)(entries / PART_SUBJOB + nsegs + 1); }
#else
__host__ __device__ inline bool part_fused_takes(uint32_t len) { return len <= PART_SUBJOB; }
__host__ __device__ inline uint32_t part_subjobs_of(uint32_t len) { return len > PART_SUBJOB ? (len + PART_SUBJOB - 1) / PART_SUBJOB : 0; }
// Upper bound of the sub-jobs of a pass: a segment that has any is longer than PART_SUBJOB, so it has fewer than 2 len / PART_SUBJOB.
inline uint32_t part_max_subjobs(uint64_t entries, uint64_t /*nsegs*/) { return (uint32_t)(2 * entries / PART_SUBJOB + 2); }
#endif

// Device scratch the grouping needs besides the two entry buffers; sizes in bytes for a plan.
struct PartScratchSizes {
  size_t matrix, partial, segs_a, segs_b, subjob_first, counts, totals;
};
inline PartScratchSizes part_scratch_sizes(const PartPlan& p) {
  PartScratchSizes s{};
  const uint64_t entries = (uint64_t)p.n * p.windows;
  s.matrix = (size_t)p.ntiles * p.nbins * 4;
  s.partial = (size_t)PART_SCAN_GROUPS * p.nbins * 4;
  uint32_t rb[4];
  const int np = part_pass_bits(p.lb, rb);
  // segment counts per level: level 1 -> nbins (or b1 merged segments when shared), then x 2^rb per pass
  uint64_t nsegs = p.shared ? (uint64_t)p.bsets * p.b1 : p.nbins, max_segs = p.nbins, max_counts = 0, max_sj = 0;
  for (int i = 0; i < np; i++) {
    const uint64_t sj = part_max_subjobs(entries, nsegs);
    max_counts = std::max<uint64_t>(max_counts, sj << rb[i]);
    max_sj = std::max<uint64_t>(max_sj, nsegs + 1);
    nsegs <<= rb[i];
    if (i + 1 < np) max_segs = std::max<uint64_t>(max_segs, nsegs);
  }
  max_sj = std::max<uint64_t>(max_sj, max_segs + 1);
  s.segs_a = s.segs_b = (size_t)max_segs * sizeof(PartSeg);
  s.subjob_first = (size_t)(max_sj + 1) * 4;
  s.counts = (size_t)max_counts * 4;
  s.totals = 64;
  return s;
}

struct PartBuffers {
  uint2* entries[2];         // each windows * n entries
  uint32_t* matrix;
  uint32_t* partial;
  PartSeg* segs[2];
  uint32_t* subjob_first;
  uint32_t* counts;
  uint32_t* totals;          // [0] = real entries (read by the accumulation), [1] = sub-jobs of the pass in flight
};


}  // namespace msm
