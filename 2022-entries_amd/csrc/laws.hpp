// laws.hpp -- the group-law policies the walking kernels (k_accumulate, k_segreduce, k_bucket_reduce) are written against.
//
//   SwLaw<E>  short Weierstrass, XYZZ accumulators, affine (x, y) base records            -- every curve (curve.hpp)
//   TeLaw<F>  twisted-Edwards image of BLS12-377 G1, extended accumulators, (X, Y, 2dXY)   -- the fast path (te.hpp)
//
// Interface:  Point (always XyzzT<T>: the 4-coordinate accumulator, so slots/buckets/fragments share one layout),
//   Base/BaseDev (what a lane gathers), set_identity, begin_run(acc) at every change of key, madd(acc, base, negate, fresh), add(acc, b) where b may be an
//   all-zero "empty" record (a bucket nobody wrote), mul_pow2(acc, k), and failed(acc): true when the law could not
//   compute the last result (TeLaw only: a vanishing denominator off the odd-order subgroup) -- kernels raise a flag then.
#pragma once
#include "curve.hpp"
#include "fp2pair.hpp"
#include "msm_types.hpp"
#include "te.hpp"

namespace msm {

// a whole record whose 64-B sectors lie `rs` bytes apart (k_accumulate_glds), as 16-byte reads
template <class B>
MSM_HD void copy_record_sectors(B& p, const unsigned char* rec, int rs) {
  constexpr int WORDS = (int)(sizeof(B) / 4), PIECES = (WORDS + 3) / 4;
  uint32_t* dst = reinterpret_cast<uint32_t*>(&p);
#pragma unroll
  for (int q = 0; q < PIECES; q++) {
    const uint4 v = *reinterpret_cast<const uint4*>(rec + (q / 4) * rs + (q % 4) * 16);
    if (4 * q < WORDS) dst[4 * q] = v.x;
    if (4 * q + 1 < WORDS) dst[4 * q + 1] = v.y;
    if (4 * q + 2 < WORDS) dst[4 * q + 2] = v.z;
    if (4 * q + 3 < WORDS) dst[4 * q + 3] = v.w;
  }
}
// one field element at the start of a 64-B sector
MSM_HD void copy_fe_sector(Fe& r, const unsigned char* src) {
  const uint4* s4 = reinterpret_cast<const uint4*>(src);
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const uint4 v = s4[q];
    r.v[4 * q] = v.x;
    r.v[4 * q + 1] = v.y;
    r.v[4 * q + 2] = v.z;
    r.v[4 * q + 3] = v.w;
  }
  const uint2 w = *reinterpret_cast<const uint2*>(src + 48);
  r.v[12] = w.x;
  r.v[13] = w.y;
}

template <class E_>
struct SwLaw {
  using E = E_;
  using T = typename E::T;
  using Md = typename E::Md;
  using MemT = T;                       // the coordinate type of the records in memory (one lane holds a whole point: LANES = 1)
  using Base = AffineT<T>;
  using BaseDev = AffineDevT<T>;
  static constexpr int LANES = 1;
  static constexpr bool CHECKS = false;
  static constexpr bool CARRY_IN = false;   // carried batches merge their chunks' buckets in a pass of their own (msm_kernels.hpp, carry_begin_run)
  static constexpr int ACC_WAVES = E::ACC_WAVES;
  static constexpr bool PREFETCH_BASE = E::PREFETCH_BASE;
  static constexpr int GATHER_SECTORS = sizeof(AffineDevT<T>) / 64;   // 64-B sectors of a record that hold data
  static constexpr int ENTRY_Q = E::ENTRY_Q;   // k_accumulate_glds: 16-byte registers of the entry queue (2 entries each); 0 = none
  static constexpr bool ITER_BARRIER = E::ITER_BARRIER;
  static MSM_HD Base from_dev(const BaseDev& d) { return d.p; }
  static MSM_HD Base from_dev_lane(const BaseDev& d, uint32_t) { return d.p; }
  static MSM_HD XyzzT<T> load_pt(const XyzzDevT<T>* p, uint32_t) { return p->p; }
  static MSM_HD void store_pt(XyzzDevT<T>* p, const XyzzT<T>& a, uint32_t) {
    XyzzDevT<T> v;
    v.p = a;
    *p = v;
  }
  static MSM_HD void set_identity(XyzzT<T>& r) { xyzz_set_inf<E>(r); }
  static MSM_HD void begin_run(XyzzT<T>&) {}   // XYZZ: the `fresh` flag makes the first madd a copy
  static MSM_HD void madd(XyzzT<T>& acc, const Base& b, bool negate, bool fresh, const Md& md) { xyzz_madd<E>(acc, b, negate, fresh, md); }
  // k_accumulate_glds: a lane's record as it lies in LDS (sector c at rec + c * rs) -> registers; madd_loaded consumes it
  static MSM_HD void load_sectors(Base& p, const unsigned char* rec, int rs, bool /*negate*/, uint32_t /*half*/) { copy_record_sectors(p, rec, rs); }
  // true = same x: the caller re-reads the base (from_dev) and calls madd_same_x
  static MSM_HD bool madd_loaded(XyzzT<T>& acc, const Base& b, bool negate, bool fresh, const Md& md) { return xyzz_madd_common<E>(acc, b, negate, fresh, md); }
  static MSM_HD void madd_same_x(XyzzT<T>& acc, const Base& b, bool negate, const Md& md) { xyzz_madd_same_x<E>(acc, b, negate, md); }
  static MSM_HD void add(XyzzT<T>& acc, const XyzzT<T>& b, const Md& md) { xyzz_add<E>(acc, b, md); }
  static MSM_HD void mul_pow2(XyzzT<T>& acc, uint32_t k, const Md& md) {
    if (xyzz_is_inf<E>(acc)) return;
    for (uint32_t i = 0; i < k; i++) xyzz_dbl<E>(acc, md);
  }
  static MSM_HD bool failed(const XyzzT<T>&) { return false; }
  static MSM_HD bool is_empty(const XyzzT<T>&) { return false; }   // XYZZ: all-zero IS the identity
  static MSM_HD bool nothing(const XyzzT<T>& a) { return xyzz_is_inf<E>(a); }   // adds nothing: an empty bucket or the identity
};

template <class F>
struct TeLaw {
  using E = FpEl<F>;
  using T = Fe;
  using Md = Modulus<F>;
  using MemT = Fe;
  using Base = TeAffine;
  using BaseDev = TeAffineDev;
  static constexpr int LANES = 1;
  static constexpr bool CHECKS = true;
  static constexpr bool CARRY_IN = true;    // later chunks of a carried batch accumulate straight onto the stored buckets
#ifndef TE_ACC_WAVES
#define TE_ACC_WAVES 2
#endif
#ifndef TE_PREFETCH
#define TE_PREFETCH true
#endif
#ifndef TE_ENTRY_Q
#define TE_ENTRY_Q 4
#endif
  static constexpr int ACC_WAVES = TE_ACC_WAVES;
  static constexpr bool PREFETCH_BASE = TE_PREFETCH;
  static constexpr int GATHER_SECTORS = 3;      // (the record may be padded to a 256-byte stride: TE_REC_PAD256, te.hpp)
  static constexpr bool ITER_BARRIER = false;
  static constexpr int ENTRY_Q = TE_ENTRY_Q;   // a whole 64-B sector of entries per refill: 140 + 16 VGPRs, still three waves per SIMD
  static MSM_HD Base from_dev(const BaseDev& d) { return d.get(); }
  static MSM_HD Base from_dev_lane(const BaseDev& d, uint32_t) { return d.get(); }
  static MSM_HD Xyzz load_pt(const XyzzDev* p, uint32_t) { return p->p; }
  static MSM_HD void store_pt(XyzzDev* p, const Xyzz& a, uint32_t) {
    XyzzDev v;
    v.p = a;
    *p = v;
  }
  static MSM_HD void set_identity(Xyzz& r) { te_set_identity<F>(r); }
  // A run's first element is added onto the identity through the same 7M formula: a cheaper "copy" branch would be taken by
  // some lane of a wave at most positions (runs are ~64 entries long), so the whole wave would pay for both paths.
  // The accumulator is reset in begin_run (a handful of moves under the run-change branch the walk has anyway).
  static MSM_HD void begin_run(Xyzz& acc) { te_set_identity<F>(acc); }
  static MSM_HD void madd(Xyzz& acc, const Base& b, bool negate, bool /*fresh*/, const Md& md) { te_madd<F>(acc, b, negate, md); }
  // k_accumulate_glds: the device record keeps one field per 64-B sector, so a negated base reads Y - X and Y + X from each
  // other's SECTOR: two LDS addresses instead of 28 selects per addition
  static MSM_HD void load_sectors(Base& p, const unsigned char* rec, int rs, bool negate, uint32_t /*half*/) {
    const int o0 = negate ? rs : 0, o1 = rs - o0;
    copy_fe_sector(p.ymx, rec + o0);
    copy_fe_sector(p.ypx, rec + o1);
    copy_fe_sector(p.td, rec + 2 * rs);
  }
  static MSM_HD bool madd_loaded(Xyzz& acc, const Base& b, bool negate, bool /*fresh*/, const Md& md) {
    te_madd<F, true>(acc, b, negate, md);
    return false;   // the law is complete: no exceptional pairs
  }
  static MSM_HD void madd_same_x(Xyzz&, const Base&, bool, const Md&) {}
  // Z = 0 never occurs in a valid point: it marks an empty (zero-filled) bucket, which adds nothing.
  static MSM_HD void add(Xyzz& acc, const Xyzz& b, const Md& md) {
    if (fe_is_zero_M<F>(b.zz)) return;
    te_add<F>(acc, b, md);
  }
  // A doubling that fails leaves Z = 0, but doubling THAT again gives Z = -(2d T^2)^2, which need not be 0: the failure is
  // frozen by zeroing the point (all-zero stays all-zero under the law), so the caller's single check after the loop sees it.
  static MSM_HD void mul_pow2(Xyzz& acc, uint32_t k, const Md& md) {
    for (uint32_t i = 0; i < k; i++) {
      te_dbl<F>(acc, md);
      if (te_failed<F>(acc)) {
        fe_zero(acc.x);
        fe_zero(acc.y);
        fe_zero(acc.zz);
        fe_zero(acc.zzz);
      }
    }
  }
  static MSM_HD bool failed(const Xyzz& a) { return te_failed<F>(a); }
  static MSM_HD bool is_empty(const Xyzz& a) { return fe_is_zero_M<F>(a.zz); }   // Z = 0 never occurs in a valid point
  static MSM_HD bool nothing(const Xyzz& a) { return fe_is_zero_M<F>(a.zz); }    // an empty bucket (the identity (0, 1, 1, 0) is added like any point)
};

#if defined(__HIPCC__)
// G2 with every Fp2 value spread over two neighbouring lanes (fp2pair.hpp): lane 2k + h of a wave holds half h (c0 / c1) of every
// coordinate of "pair" k's points.  The records in memory are those of SwLaw<Fp2El<F, NB>> -- a kernel of either form reads
// what the other wrote -- and the group law is curve.hpp's, instantiated over the paired coordinate policy.
template <class F, int NB>
struct SwPairLaw {
  using E = Fp2PairEl<F, NB>;
  using T = Fe;
  using MemT = Fe2;
  using Md = typename E::Md;
  using Base = AffineT<Fe>;
  using BaseDev = AffineDevT<Fe2>;
  static constexpr int LANES = 2;
  static constexpr bool CHECKS = false;
  static constexpr bool CARRY_IN = false;
#ifndef MSM_G2P_ACC_WAVES
#define MSM_G2P_ACC_WAVES 2
#endif
#ifndef MSM_G2P_ENTRY_Q
#define MSM_G2P_ENTRY_Q 2
#endif
  static constexpr int ACC_WAVES = MSM_G2P_ACC_WAVES;
  static constexpr bool PREFETCH_BASE = true;
  static constexpr int GATHER_SECTORS = 4;      // 224 bytes of data in a 256-byte record
  static constexpr int ENTRY_Q = MSM_G2P_ENTRY_Q;
  static constexpr bool ITER_BARRIER = false;
  static __device__ __forceinline__ Base from_dev_lane(const BaseDev& d, uint32_t h) {
    Base b;
    b.x = fe_load8(pair_coord(&d, 0, h));
    b.y = fe_load8(pair_coord(&d, 1, h));
    return b;
  }
  static __device__ __forceinline__ XyzzT<Fe> load_pt(const XyzzDevT<Fe2>* p, uint32_t h) {
    XyzzT<Fe> r;
    r.x = fe_load8(pair_coord(p, 0, h));
    r.y = fe_load8(pair_coord(p, 1, h));
    r.zz = fe_load8(pair_coord(p, 2, h));
    r.zzz = fe_load8(pair_coord(p, 3, h));
    return r;
  }
  static __device__ __forceinline__ void store_pt(XyzzDevT<Fe2>* p, const XyzzT<Fe>& a, uint32_t h) {
    fe_store8(pair_coord(p, 0, h), a.x);
    fe_store8(pair_coord(p, 1, h), a.y);
    fe_store8(pair_coord(p, 2, h), a.zz);
    fe_store8(pair_coord(p, 3, h), a.zzz);
  }
  // k_accumulate_glds: the record's 64-B sectors lie `rs` bytes apart in LDS; half h of x starts at byte 56 h of the record, of y at
  // 112 + 56 h.  Every 8-byte unit lies inside one sector.
  static __device__ __forceinline__ void load_sectors(Base& p, const unsigned char* rec, int rs, bool /*negate*/, uint32_t h) {
#pragma unroll
    for (int k = 0; k < NL / 2; k++) {
      const uint32_t ox = 56u * h + 8u * k, oy = 112u + 56u * h + 8u * k;
      const uint2 vx = *reinterpret_cast<const uint2*>(rec + (ox >> 6) * rs + (ox & 63u));
      const uint2 vy = *reinterpret_cast<const uint2*>(rec + (oy >> 6) * rs + (oy & 63u));
      p.x.v[2 * k] = vx.x;
      p.x.v[2 * k + 1] = vx.y;
      p.y.v[2 * k] = vy.x;
      p.y.v[2 * k + 1] = vy.y;
    }
  }
  static __device__ __forceinline__ void set_identity(XyzzT<Fe>& r) { xyzz_set_inf<E>(r); }
  static __device__ __forceinline__ void begin_run(XyzzT<Fe>&) {}
  static __device__ __forceinline__ bool madd_loaded(XyzzT<Fe>& acc, const Base& b, bool negate, bool fresh, const Md& md) { return xyzz_madd_common<E>(acc, b, negate, fresh, md); }
  static __device__ __forceinline__ void madd_same_x(XyzzT<Fe>& acc, const Base& b, bool negate, const Md& md) { xyzz_madd_same_x<E>(acc, b, negate, md); }
  static __device__ __forceinline__ void add(XyzzT<Fe>& acc, const XyzzT<Fe>& b, const Md& md) { xyzz_add<E>(acc, b, md); }
  static __device__ __forceinline__ void mul_pow2(XyzzT<Fe>& acc, uint32_t k, const Md& md) {
    if (xyzz_is_inf<E>(acc)) return;
    for (uint32_t i = 0; i < k; i++) xyzz_dbl<E>(acc, md);
  }
  static __device__ __forceinline__ bool failed(const XyzzT<Fe>&) { return false; }
  static __device__ __forceinline__ bool is_empty(const XyzzT<Fe>&) { return false; }
  static __device__ __forceinline__ bool nothing(const XyzzT<Fe>& a) { return xyzz_is_inf<E>(a); }
};
#endif

}  // namespace msm
