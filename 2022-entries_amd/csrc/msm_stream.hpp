// msm_stream.hpp -- arkworks' streaming MSM accumulators over the MI355X engine (included by msm_engine.hip).
//
// Reference: ARK ec/src/msm/variable_base/stream_pippenger.rs -- the two accumulators a prover holds across rounds:
//   ChunkedPippenger  (:11-75)   buffer (base, BigInt scalar) pairs; when the buffer holds buf_size pairs,
//                                result += msm_bigint(buffer), clear;  finalize() flushes what is left and returns result.
//   HashMapPippenger  (:78-140)  a map base -> Fr: a pair whose base is already in the map ADDS its scalar to that entry (field
//                                addition mod r); when the map holds buf_size DISTINCT bases, result += msm_bigint(keys,
//                                values.into_bigint()), clear;  finalize() as above.
// Here the flush is the engine's stateless pipeline (msm_stateless.hpp): the buffered pairs stream to the device in slices
// while earlier slices compute; the running result stays on the host as a normalised Projective image and partial sums are
// added with the host group law (mi355_msm_fold).  `add` takes any number of pairs and behaves exactly like that many
// single-pair adds (the buffer is flushed the moment it reaches buf_size).  finalize() returns the result and resets the
// accumulator to empty -- arkworks' finalize consumes self; a C object has to say what happens next.
#pragma once

#include <unordered_map>

struct mi355_msm_stream {
  int curve = 0;
  int device = -1;
  bool hashmap = false;
  size_t buf_size = 0;
  size_t stride = 0;                   // packed Affine image: 104 (G1) / 200 (G2)
  std::vector<uint8_t> bases, scalars; // the buffered pairs, `stride` and 32 bytes each
  std::unordered_map<std::string, size_t> index;   // hashmap mode: affine image -> position in the buffers
  std::vector<uint8_t> result;         // normalised Projective image of everything flushed so far
  long opt_scalars_montgomery = 0, opt_window_bits = 0;
  uint64_t flushes = 0, merged = 0;
};

namespace {

struct U256 {
  uint64_t w[4];
};
inline U256 u256_load(const uint8_t* p) {
  U256 v;
  memcpy(v.w, p, 32);
  return v;
}
inline bool u256_geq(const U256& a, const U256& b) {
  for (int i = 3; i >= 0; i--)
    if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
  return true;
}
inline void u256_sub(U256& a, const U256& b) {
  unsigned __int128 borrow = 0;
  for (int i = 0; i < 4; i++) {
    const unsigned __int128 d = (unsigned __int128)a.w[i] - b.w[i] - (uint64_t)borrow;
    a.w[i] = (uint64_t)d;
    borrow = (d >> 64) & 1;
  }
}
// (a + b) mod r for a, b < 2^256 (inputs are reduced first; r > 2^252 for both curves, so a handful of subtractions at most)
inline U256 fr_add(U256 a, U256 b, const U256& r) {
  while (u256_geq(a, r)) u256_sub(a, r);
  while (u256_geq(b, r)) u256_sub(b, r);
  unsigned __int128 carry = 0;
  U256 s;
  for (int i = 0; i < 4; i++) {
    const unsigned __int128 t = (unsigned __int128)a.w[i] + b.w[i] + (uint64_t)carry;
    s.w[i] = (uint64_t)t;
    carry = t >> 64;
  }
  if (carry || u256_geq(s, r)) u256_sub(s, r);   // a + b < 2r < 2^255: no carry out in practice
  return s;
}
inline U256 fr_modulus(int curve) {
  const uint32_t* r = is_381(curve) ? msm::Bls12_381_Fr::R : msm::Bls12_377_Fr::R;
  U256 m;
  for (int i = 0; i < 4; i++) m.w[i] = (uint64_t)r[2 * i] | ((uint64_t)r[2 * i + 1] << 32);
  return m;
}

void stream_reset_result(mi355_msm_stream* s) {
  const size_t cb = coord_bytes(s->curve);
  s->result.assign(3 * cb, 0);
  // (1, 1, 0) in Montgomery form is what the engine writes for the point at infinity: take it from the fold of nothing
  take(mi355_msm_fold(s->curve, s->result.data(), s->result.data(), 0));
}

// result += MSM(buffer); buffer cleared.
void stream_flush(mi355_msm_stream* s) {
  const size_t n = s->scalars.size() / 32;
  if (n == 0) return;
  const size_t pb = 3 * coord_bytes(s->curve);
  std::vector<uint8_t> two(2 * pb);
  memcpy(two.data(), s->result.data(), pb);
  {
    StatelessLease ws(s->curve, s->device);
    // through set_option (its validation, its reset of the fitted chunk), and back to the defaults on every way out: the context
    // returns to the pool of the plain stateless call
    struct Restore {
      mi355_msm_ctx* c;
      ~Restore() {
        for (const char* k : {"scalars_montgomery", "window_bits", "assume_subgroup"}) {
          RustError e = mi355_msm_set_option(c, k, 0);
          if (e.message) free(e.message);
        }
      }
    } restore{ws.ctx};
    take(mi355_msm_set_option(ws.ctx, "scalars_montgomery", s->opt_scalars_montgomery));
    take(mi355_msm_set_option(ws.ctx, "window_bits", s->opt_window_bits));
    // (a pooled context keeps whatever the last stateless call set: the accumulators are exact for any curve point unless the
    //  environment says the bases are in the subgroup, as for mi355_msm -- ADVICE r4)
    {
      const char* sub_env = getenv("MI355_MSM_ASSUME_SUBGROUP");
      take(mi355_msm_set_option(ws.ctx, "assume_subgroup", (sub_env && *sub_env && atol(sub_env) != 0) ? 1 : 0));
    }
    stateless_run(ws.ctx, two.data() + pb, s->bases.data(), n, s->scalars.data(), s->stride);
    ws.keep();
  }
  take(mi355_msm_fold(s->curve, s->result.data(), two.data(), 2));
  s->bases.clear();
  s->scalars.clear();
  s->index.clear();
  s->flushes++;
}

void stream_add_one(mi355_msm_stream* s, const uint8_t* base, const uint8_t* scalar, const U256& r) {
  const size_t cb2 = 2 * coord_bytes(s->curve);
  if (s->hashmap) {
    // the key is what arkworks hashes: x, y and the infinity flag (the padding behind the flag is not part of the point)
    std::string key((const char*)base, cb2);
    key.push_back(base[cb2] ? 1 : 0);
    auto it = s->index.find(key);
    if (it != s->index.end()) {
      uint8_t* slot = s->scalars.data() + it->second * 32;
      const U256 sum = fr_add(u256_load(slot), u256_load(scalar), r);
      memcpy(slot, sum.w, 32);
      s->merged++;
      return;   // the map did not grow: no flush check (stream_pippenger.rs:112-116 tests len() after every add; it is unchanged)
    }
    s->index.emplace(std::move(key), s->scalars.size() / 32);
  }
  const size_t at = s->bases.size();
  s->bases.resize(at + s->stride, 0);
  memcpy(s->bases.data() + at, base, cb2 + 1);
  s->scalars.insert(s->scalars.end(), scalar, scalar + 32);
  if (s->scalars.size() / 32 == s->buf_size) stream_flush(s);
}

}  // namespace

extern "C" {

RustError mi355_msm_stream_create(mi355_msm_stream** out, int curve, int device, size_t max_msm_buffer, int hashmap) {
  return guarded_dev([&] {
    if (!out) bad_arg("null stream out-pointer");
    *out = nullptr;
    if (!known_curve(curve)) bad_arg("unknown curve id %d", curve);
    if (max_msm_buffer == 0 || max_msm_buffer >= (1ull << 31)) bad_arg("max_msm_buffer %zu out of range [1, 2^31)", max_msm_buffer);
    // fail now, not at the first flush, when there is no device to flush to (no CPU fallback)
    const int count = require_device();
    if (device >= count) bad_arg("device %d out of range (%d visible)", device, count);
    if (device < 0) HIP_OK(hipGetDevice(&device));
    mi355_msm_stream* s = new mi355_msm_stream();
    s->curve = curve;
    s->device = device;
    s->hashmap = hashmap != 0;
    s->buf_size = max_msm_buffer;
    s->stride = 2 * coord_bytes(curve) + 8;
    try {
      stream_reset_result(s);
    } catch (...) {
      delete s;
      throw;
    }
    *out = s;
  });
}

RustError mi355_msm_stream_set_option(mi355_msm_stream* s, const char* key, long value) {
  return guarded([&] {
    if (!s || !key) bad_arg("null argument");
    const std::string k(key);
    if (k == "scalars_montgomery")
      s->opt_scalars_montgomery = value != 0;
    else if (k == "window_bits") {
      if (value != 0 && (value < 2 || value > 24)) bad_arg("window_bits %ld out of range [2, 24]", value);
      s->opt_window_bits = value;
    } else
      bad_arg("unknown stream option '%s'", key);
  });
}

RustError mi355_msm_stream_add(mi355_msm_stream* s, const void* affine, size_t stride, const void* scalars, size_t count) {
  return guarded_dev([&] {
    if (!s) bad_arg("null stream");
    if (count && (!affine || !scalars)) bad_arg("null bases or scalars pointer");
    if (count && stride < 2 * coord_bytes(s->curve) + 1) bad_arg("affine stride %zu too small", stride);
    const U256 r = fr_modulus(s->curve);
    for (size_t i = 0; i < count; i++) stream_add_one(s, (const uint8_t*)affine + i * stride, (const uint8_t*)scalars + i * 32, r);
  });
}

RustError mi355_msm_stream_finalize(mi355_msm_stream* s, void* out_projective) {
  return guarded_dev([&] {
    if (!s || !out_projective) bad_arg("null argument");
    stream_flush(s);
    memcpy(out_projective, s->result.data(), s->result.size());
    stream_reset_result(s);
  });
}

RustError mi355_msm_stream_query(mi355_msm_stream* s, const char* key, uint64_t* value) {
  return guarded([&] {
    if (!s || !key || !value) bad_arg("null argument");
    const std::string k(key);
    if (k == "buffered")
      *value = s->scalars.size() / 32;
    else if (k == "flushes")
      *value = s->flushes;
    else if (k == "merged")
      *value = s->merged;
    else if (k == "buf_size")
      *value = s->buf_size;
    else
      bad_arg("unknown stream query '%s'", key);
  });
}

RustError mi355_msm_stream_destroy(mi355_msm_stream* s) {
  return guarded([&] { delete s; });
}

}  // extern "C"
