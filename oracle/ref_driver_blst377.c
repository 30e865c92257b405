/* ref_driver_blst377.c -- thin C driver around the blst 0.3.10 copy the REFERENCE holds under
 * team-division/prize1-marlin-verifier/Jackytan2018/external/blst-0.3.10/blst, compiled from where it lies by oracle/Makefile
 * into oracle/_ref/libblst377.so together with blst's src/server.c and build/assembly.S (plain gcc, one line; no reference build
 * system, no stand-ins).  No reference source is copied: the header is #included by path.  TEST INFRASTRUCTURE ONLY.
 *
 * What that copy IS (read before trusting a name): its author re-targeted blst to the BLS12-377 base field -- src/consts.c
 * carries p = 0x01ae3a46...0001, p0, RR and the BLS12-377 group order under the BLS12_381_* names (consts.c:27-35, 42, 56-61,
 * 90-93; the 381 values are left behind as comments).  Checked here: blst_fp_mul/add/sqr are exact modulo the BLS12-377 p; the
 * generators and the inversion (fixed addition chains for the 381 exponent) are not usable, and are not used.  So:
 *
 *   blst_p1s_mult_pippenger   bindings/blst.h:238, src/multi_scalar.c:264-402: Booth-recoded signed windows over XYZZ buckets
 *                             (an algorithm unrelated to arkworks') -- a reference-computed **BLS12-377 G1** MSM: the a = 0
 *                             addition/doubling formulas never touch the curve constant b.
 *   blst_p2s_mult_pippenger   bindings/blst.h:262, over src/e2.c and blst's Fp2 tower, which is hard-wired to u^2 = -1.  Over the
 *                             BLS12-377 prime that quotient is a RING, Fp[u]/(u^2 + 1) = Fp x Fp (p = 1 mod 4), not BLS12-377's
 *                             Fq2 (u^2 = -5).  The group law is a set of polynomial identities, so on points of ONE curve
 *                             y^2 = x^3 + b over that ring (componentwise a pair of curves over Fp) the reference's G2 Pippenger
 *                             and the oracle's Fp2 template instance with beta = -1 (msm_oracle.c curve id 4) must agree: a
 *                             reference computation of the G2-shaped code path (Fp2 Karatsuba, XYZZ over Fp2, Booth windows).
 *
 * Layouts: blst_fp is 6 x u64 in Montgomery form with R = 2^384 -- the image arkworks' Fq has -- so an arkworks
 * Affine record (x | y | infinity byte | pad) maps to blst_p{1,2}_affine by copying x | y; blst marks an affine point at infinity
 * by all-zero coordinates.  Results are the RAW Jacobian triple X | Y | Z blst returns (its to_affine needs the inversion): the
 * caller normalises with oracle_jac_normalize. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "blst.h"

/* sum k_i P_i with the reference's G2 Pippenger.  bases: 200-byte Affine images over the coordinate ring (flag at byte 192);
 * scalars: 32-byte LE integers < 2^nbits; out288 = X | Y | Z (Jacobian, not normalised).  Returns 1 when Z = 0. */
int refblst_g2_msm(const uint8_t* bases, size_t stride, const uint8_t* scalars, size_t n, size_t nbits, uint8_t* out288) {
  blst_p2 ret;
  memset(&ret, 0, sizeof ret);
  if (n >= 1) {
    /* (the tile loop reads one scalar ahead: a single pair is run as two, the second the point at infinity times zero) */
    const size_t m = n < 2 ? 2 : n;
    blst_p2_affine* pts = (blst_p2_affine*)calloc(m, sizeof *pts);
    uint8_t* sc = (uint8_t*)calloc(m * 32 + 64, 1);
    limb_t* scratch = (limb_t*)malloc(blst_p2s_mult_pippenger_scratch_sizeof(m));
    if (!pts || !sc || !scratch) return -1;
    for (size_t i = 0; i < n; i++)
      if (!bases[i * stride + 192]) memcpy(&pts[i], bases + i * stride, 192);
    memcpy(sc, scalars, n * 32);
    const blst_p2_affine* pp[2] = {pts, NULL};
    const byte* ss[2] = {sc, NULL};
    blst_p2s_mult_pippenger(&ret, pp, m, ss, nbits, scratch);
    free(pts);
    free(sc);
    free(scratch);
  }
  memcpy(out288, &ret, 288);
  return blst_p2_is_inf(&ret) ? 1 : 0;
}

/* the same over G1 (104-byte Affine images, flag at byte 96); out144 = X | Y | Z (Jacobian, not normalised). */
int refblst_g1_msm(const uint8_t* bases, size_t stride, const uint8_t* scalars, size_t n, size_t nbits, uint8_t* out144) {
  blst_p1 ret;
  memset(&ret, 0, sizeof ret);
  if (n >= 1) {
    const size_t m = n < 2 ? 2 : n;
    blst_p1_affine* pts = (blst_p1_affine*)calloc(m, sizeof *pts);
    uint8_t* sc = (uint8_t*)calloc(m * 32 + 64, 1);
    limb_t* scratch = (limb_t*)malloc(blst_p1s_mult_pippenger_scratch_sizeof(m));
    if (!pts || !sc || !scratch) return -1;
    for (size_t i = 0; i < n; i++)
      if (!bases[i * stride + 96]) memcpy(&pts[i], bases + i * stride, 96);
    memcpy(sc, scalars, n * 32);
    const blst_p1_affine* pp[2] = {pts, NULL};
    const byte* ss[2] = {sc, NULL};
    blst_p1s_mult_pippenger(&ret, pp, m, ss, nbits, scratch);
    free(pts);
    free(sc);
    free(scratch);
  }
  memcpy(out144, &ret, 144);
  return blst_p1_is_inf(&ret) ? 1 : 0;
}

/* field products through blst's own arithmetic: Fp (48-byte Montgomery images) and its Fp2 tower (c0 | c1, u^2 = -1). */
void refblst_fp_mul(const uint8_t* a48, const uint8_t* b48, uint8_t* out48) {
  blst_fp a, b, r;
  memcpy(&a, a48, 48);
  memcpy(&b, b48, 48);
  blst_fp_mul(&r, &a, &b);
  memcpy(out48, &r, 48);
}
void refblst_fp2_mul(const uint8_t* a96, const uint8_t* b96, uint8_t* out96) {
  blst_fp2 a, b, r;
  memcpy(&a, a96, 96);
  memcpy(&b, b96, 96);
  blst_fp2_mul(&r, &a, &b);
  memcpy(out96, &r, 96);
}
