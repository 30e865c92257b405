// msm_types.hpp -- device data layouts shared by the kernels and the host orchestration.
#pragma once
#include "curve.hpp"

namespace msm {

constexpr uint32_t KEY_NONE = 0xffffffffu;
constexpr uint32_t IDX_MASK = 0x7fffffffu;

// Device layouts: AoS, 16-byte aligned so one lane's gather/store is a run of dwordx4 accesses.
//   G1: affine 112 B (2 x 14 limbs), XYZZ 224 B;   G2: affine 224 B, XYZZ 448 B.
template <class T>
struct alignas(128) AffineDevT {
  AffineT<T> p;
};
template <class T>
struct alignas(16) XyzzDevT {
  XyzzT<T> p;
};
using AffineDev = AffineDevT<Fe>;
using XyzzDev = XyzzDevT<Fe>;
static_assert(sizeof(AffineDev) == 128 && sizeof(AffineDevT<Fe2>) == 256, "device affine layout");
static_assert(sizeof(XyzzDev) == 224 && sizeof(XyzzDevT<Fe2>) == 448, "device xyzz layout");

template <class T>
struct SegOutT {
  XyzzDevT<T>* buckets;
  XyzzDevT<T>* slots;   // 2 per lane: [2t] head, [2t+1] tail
  uint32_t* slot_keys;  // KEY_NONE = empty slot
  // 1: the buckets already hold the sums of the batch's earlier chunks (carried buckets): the run that STARTS a bucket in this chunk
  // begins from the stored value instead of the identity, so the chunk's sums land on top of it with no merge pass
  uint32_t carry_in = 0;
};
using SegOut = SegOutT<Fe>;

}  // namespace msm
