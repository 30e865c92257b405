// ref_driver_377.cpp -- thin extern "C" driver around the REFERENCE's own host-side BLS12-377 field and XYZZ
// curve code (CMB yrrid-ff-ec/HostCurve.cpp), compiled from where it lies under /root/reference by
// oracle/Makefile into oracle/_ref/libref377.so.  No reference source is copied: the file is #included by
// path at build time.  TEST INFRASTRUCTURE ONLY -- it pins oracle/msm_oracle.c and pymodel.py to the
// reference's arithmetic (field mul, XYZZ add/dbl/normalize) for BLS12-377.
#include <stdint.h>
#include <string.h>

#include "yrrid-ff-ec/HostCurve.cpp"

typedef Host::BLS12377::G1Montgomery Field;
typedef Host::PointXYZZ<Field> PointXYZZ;
typedef Host::AccumulatorXYZZ<Field> AccumulatorXYZZ;

extern "C" {

// out = a * b * 2^-384 mod p on 12 x u32 Montgomery words, canonical.
void ref377_fp_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  Field::Value x, y, z;
  Field::load(x, (uint32_t*)a);
  Field::load(y, (uint32_t*)b);
  Field::mul(z, x, y);
  Field::reduce(z, z);
  Field::store(out, z);
}

// sum k_i * P_i with the reference's XYZZ add/dbl (double-and-add per point), normalised by the reference's
// PointXYZZ::normalize.  Writes x | y | z (48 B each; z = R, or all-zero triple when the sum is infinity)
// and returns 1 when the result is infinity.
int ref377_msm_naive(const uint8_t* bases, size_t stride, const uint8_t* scalars, size_t n, uint8_t* out144) {
  AccumulatorXYZZ total;
  for (size_t i = 0; i < n; i++) {
    const uint8_t* b = bases + i * stride;
    if (b[96]) continue;
    uint32_t w[48];
    memcpy(w, b, 96);
    Field::Value r;
    Field::setR(r);
    Field::store(w + 24, r);
    Field::store(w + 36, r);
    PointXYZZ p;
    p.load(w);
    AccumulatorXYZZ acc;
    uint64_t k[4];
    memcpy(k, scalars + 32 * i, 32);
    for (int bit = 255; bit >= 0; bit--) {
      PointXYZZ cur = acc.xyzz;
      acc.dbl(cur);
      if ((k[bit >> 6] >> (bit & 63)) & 1) acc.add(p);
    }
    total.add(acc.xyzz);
  }
  PointXYZZ res = total.xyzz;
  if (Field::isZero(res.zz)) {
    memset(out144, 0, 144);
    return 1;
  }
  res.normalize();
  res.reduce();
  uint32_t w[48];
  res.store(w);
  memcpy(out144, w, 96);        // x, y
  memcpy(out144 + 96, w + 24, 48);  // zz = R after normalize
  return 0;
}
}
