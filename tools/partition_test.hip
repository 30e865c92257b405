// partition_test.hip -- standalone check + timing of the bucket-grouping kernels (csrc/partition.hpp) against a CPU model:
// every (key, value) entry the digits of the scalars define must come out exactly once, keys non-decreasing, nothing else.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/partition_test.hip -o tools/partition_test
//   tools/partition_test [npow=20] [c=0 (auto)] [levels=0 (no tables) | 1 (a table level per window) | k (k levels: windows g, g + G, ...
//                        share bucket set g)] [pattern=0 uniform|1 all-equal|2 small|3 zeros+ones] [reps=3]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#include "../2022-entries_amd/csrc/partition.hpp"

using namespace msm;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

struct Agg { uint64_t count = 0, sum = 0, x = 0; };

int main(int argc, char** argv) {
  const int npow = argc > 1 ? atoi(argv[1]) : 20;
  uint32_t c = argc > 2 ? atoi(argv[2]) : 0;
  const int levels_arg = argc > 3 ? atoi(argv[3]) : 0;
  const bool shared = levels_arg != 0;
  const int pattern = argc > 4 ? atoi(argv[4]) : 0;
  const int reps = argc > 5 ? atoi(argv[5]) : 3;
  const uint32_t n = npow >= 0 ? (1u << npow) : (uint32_t)(-npow);   // negative: a literal (ragged) count
  if (!c) c = npow >= 24 ? 20 : (npow >= 16 ? 14 : 9);
  const uint32_t windows = (257 + c - 1) / c;
  uint32_t levels = !shared ? 1 : (levels_arg == 1 ? windows : std::min<uint32_t>((uint32_t)levels_arg, windows));
  const uint32_t bsets = (windows + levels - 1) / levels;
  levels = (windows + bsets - 1) / bsets;
  const uint32_t table_stride = shared ? n : 0;
  const uint32_t idx0 = shared ? 0 : 5;   // a chunk offset
  const size_t nbases = shared ? (size_t)levels * n : n + idx0;
  printf("n=%u c=%u windows=%u levels=%u bucket_sets=%u pattern=%d\n", n, c, windows, levels, bsets, pattern);

  std::mt19937_64 rng(1234 + npow + pattern);
  std::vector<uint64_t> sc((size_t)n * 4);
  for (uint32_t i = 0; i < n; i++) {
    uint64_t* s = &sc[(size_t)i * 4];
    switch (pattern) {
      case 0: s[0] = rng(); s[1] = rng(); s[2] = rng(); s[3] = rng() & 0x0fffffffffffffffull; break;
      case 1: s[0] = 0x123456789abcdef1ull; s[1] = 0xfedcba9876543210ull; s[2] = 0x0f0f0f0f0f0f0f0full; s[3] = 0x0123456789abcdefull; break;
      case 2: s[0] = rng(); s[1] = s[2] = s[3] = 0; break;
      default: s[0] = rng() & 1; s[1] = s[2] = s[3] = 0; if ((i & 15) == 3) s[0] = ~0ull, s[1] = ~0ull, s[2] = ~0ull, s[3] = ~0ull; break;
    }
  }
  std::vector<uint8_t> inf(nbases, 0);
  for (size_t i = 7; i < nbases; i += 1001) inf[i] = 1;

  // CPU model (the rule of k_digits / CMB ProcessSignedDigits.cu:118-151)
  std::map<uint32_t, Agg> want;
  uint64_t want_total = 0;
  const bool model = n <= (1u << 22);
  if (model) {
    const uint32_t half = 1u << (c - 1);
    for (uint32_t i = 0; i < n; i++) {
      uint32_t s[8];
      memcpy(s, &sc[(size_t)i * 4], 32);
      uint32_t carry = 0;
      for (uint32_t w = 0; w < windows; w++) {
        uint32_t v = (s[0] & ((1u << c) - 1)) + carry;
        for (int j = 0; j < 7; j++) s[j] = (s[j] >> c) | (s[j + 1] << (32 - c));
        s[7] >>= c;
        const bool neg = v > half;
        const uint32_t d = neg ? (1u << c) - v : v;
        carry = neg;
        const uint32_t idx = idx0 + i + (w / bsets) * table_stride;
        if (d == 0 || inf[idx]) continue;
        const uint32_t key = (w % bsets) * half + d - 1;
        const uint32_t val = idx | (neg ? 0x80000000u : 0);
        Agg& a = want[key];
        a.count++; a.sum += val; a.x ^= (uint64_t)val * 0x9e3779b97f4a7c15ull;
        want_total++;
      }
    }
  }

  const PartPlan p = part_plan(n, c, windows, shared ? levels : 1, idx0, table_stride);
  const PartScratchSizes sz = part_scratch_sizes(p);
  const uint64_t E = (uint64_t)n * windows;
  printf("hb=%u lb=%u nbins=%u ntiles=%u  scratch: matrix %.1f MB counts %.1f MB segs %.1f MB\n", p.hb, p.lb, p.nbins, p.ntiles,
         sz.matrix / 1e6, sz.counts / 1e6, sz.segs_a / 1e6);
  uint32_t* d_sc; uint8_t* d_inf;
  CHECK(hipMalloc(&d_sc, (size_t)n * 32)); CHECK(hipMalloc(&d_inf, nbases));
  CHECK(hipMemcpy(d_sc, sc.data(), (size_t)n * 32, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_inf, inf.data(), nbases, hipMemcpyHostToDevice));
  PartBuffers b{};
  CHECK(hipMalloc(&b.entries[0], E * 8 + 64)); CHECK(hipMalloc(&b.entries[1], E * 8 + 64));
  CHECK(hipMalloc(&b.matrix, sz.matrix)); CHECK(hipMalloc(&b.partial, sz.partial));
  CHECK(hipMalloc(&b.segs[0], sz.segs_a)); CHECK(hipMalloc(&b.segs[1], sz.segs_b));
  CHECK(hipMalloc(&b.subjob_first, sz.subjob_first)); CHECK(hipMalloc(&b.counts, sz.counts)); CHECK(hipMalloc(&b.totals, sz.totals));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1, em; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&em));
  PartProbe probe;
  for (int i = 0; i <= PartProbe::MAX; i++) CHECK(hipEventCreate(&probe.ev[i]));
  float kbest[PartProbe::MAX];
  for (float& v : kbest) v = 1e30f;
  int res = 0;
  float best = 1e30f, best_l1 = 0;
  for (int r = 0; r < reps; r++) {
    CHECK(hipMemsetAsync(b.entries[0], 0xff, E * 8, st));
    CHECK(hipMemsetAsync(b.entries[1], 0xff, E * 8, st));
    CHECK(hipEventRecord(e0, st));
    hipError_t err;
    res = part_run<Bls12_377_Fr, false>(d_sc, d_inf, p, b, st, em, err, &probe);
    CHECK(err);
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    for (int i = 0; i < probe.n; i++) {
      float k;
      CHECK(hipEventElapsedTime(&k, probe.ev[i], probe.ev[i + 1]));
      if (k < kbest[i]) kbest[i] = k;
    }
    float ms, ms1; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipEventElapsedTime(&ms1, e0, em));
    if (ms < best) { best = ms; best_l1 = ms1; }
  }
  uint32_t totals[2];
  CHECK(hipMemcpy(totals, b.totals, 8, hipMemcpyDeviceToHost));
  printf("grouping: %.3f ms (level 1 incl. histogram + scan %.3f ms), %u real entries of %llu (%.2f G entries/s)\n", best, best_l1, totals[0],
         (unsigned long long)E, E / best / 1e6);
  printf("generic sub-jobs of the last pass: %u\n", totals[1]);
  printf("per kernel (best of %d, ms):", reps);
  for (int i = 0; i < probe.n; i++) printf("  %s %.3f", probe.name[i], kbest[i]);
  printf("\n");
  int bad = 0;
  if (model) {
    std::vector<uint2> out(totals[0]);
    CHECK(hipMemcpy(out.data(), b.entries[res], (size_t)totals[0] * 8, hipMemcpyDeviceToHost));
    if (totals[0] != want_total) { printf("FAIL: %u entries, model has %llu\n", totals[0], (unsigned long long)want_total); bad++; }
    std::map<uint32_t, Agg> got;
    uint32_t prev = 0;
    for (size_t i = 0; i < out.size(); i++) {
      const uint32_t key = out[i].y, val = out[i].x;
      if (key < prev) { if (bad < 5) printf("FAIL: keys not sorted at %zu (%u after %u)\n", i, key, prev); bad++; }
      prev = key;
      Agg& a = got[key];
      a.count++; a.sum += val; a.x ^= (uint64_t)val * 0x9e3779b97f4a7c15ull;
    }
    if (got.size() != want.size()) { printf("FAIL: %zu distinct keys, model has %zu\n", got.size(), want.size()); bad++; }
    for (auto& kv : want) {
      auto it = got.find(kv.first);
      if (it == got.end() || it->second.count != kv.second.count || it->second.sum != kv.second.sum || it->second.x != kv.second.x) {
        if (bad < 8) printf("FAIL: key %u differs (count %llu vs %llu)\n", kv.first, it == got.end() ? 0ull : (unsigned long long)it->second.count,
                            (unsigned long long)kv.second.count);
        bad++;
      }
    }
    printf(bad ? "RESULT: FAIL (%d problems)\n" : "RESULT: ok (%zu keys, every entry accounted for)\n", bad ? bad : 0, want.size());
  } else {
    printf("RESULT: timing only (n too large for the CPU model)\n");
  }
  return bad ? 1 : 0;
}
