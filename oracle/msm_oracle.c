/* msm_oracle.c -- CPU restatement of the reference MSM path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the reported CPU baseline.  The product path (2022-entries_amd/) never links or calls it.
 *
 * It restates, in plain C, the algorithm of arkworks' VariableBaseMSM (the oracle the reference tests
 * compare against, P1A combined-top-solutions/tests/msm.rs:28-38).  arkworks is Rust and cannot be built
 * in this image (no cargo), so each function cites the reference lines it follows
 * (ARK = open-division/prize4-msm-wasm/snarkify/zprize-prize4-15ac8c55-arkworks-algebra):
 *
 *   fp_add/sub/dbl/mul      ARK ff/src/fields/models/fp/montgomery_backend.rs:108-136, 146-201 (CIOS, 64-bit limbs)
 *   jac_double              ARK ec/src/models/short_weierstrass.rs:815-848   (dbl-2009-l, a = 0)
 *   jac_add_mixed           ARK ec/src/models/short_weierstrass.rs:886-948   (madd-2007-bl, doubles when equal)
 *   jac_add                 ARK ec/src/models/short_weierstrass.rs:979-1040  (add-2007-bl)
 *   jac_to_affine           ARK ec/src/models/short_weierstrass.rs:1093-1115
 *   oracle_msm              ARK ec/src/msm/variable_base/mod.rs:68-162 with the window rule of
 *                           ARK ec/src/msm/mod.rs:54-57; one thread per window like the rayon `cfg_into_iter!`
 *
 * PARITY PIN: this restatement is pinned (tests/test_oracle.py) against (i) the literal constants and
 * edge-case points the reference holds, (ii) oracle/_ref: the reference's own C/C++ sources compiled here --
 * yrrid's host XYZZ curve code for BLS12-377 (CMB yrrid-ff-ec/HostCurve.cpp) and yrrid's C BLS12-381 MSM
 * (open-division/prize4-msm-wasm/yrrid/C/MSM.c) -- and (iii) the independent Python model (pymodel.py).
 * The reference stores no MSM output vectors.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NLIMB 6
typedef unsigned __int128 u128;

typedef struct {
  uint64_t l[NLIMB];
} fp_t;

typedef struct {
  fp_t p;        /* modulus */
  uint64_t inv;  /* -p^-1 mod 2^64 */
  fp_t one;      /* R mod p, R = 2^384 */
  fp_t r2;       /* R^2 mod p */
  int scalar_bits;
} field_t;

typedef struct {
  fp_t x, y, z;
} jac_t;

typedef struct {
  fp_t x, y;
  int inf;
} aff_t;

/* Moduli: ARKC bls12_377/src/fields/fq.rs:4, bls12_381/src/fields/fq.rs:4 (same limbs as SPK ff/bls12-377.hpp:10-14,
 * ff/bls12-381.hpp:10-14).  Scalar bit sizes: fr.rs:24 (253 bits), bls12_381 fr.rs:4 (255 bits). */
static const uint64_t P377[6] = {0x8508c00000000001ull, 0x170b5d4430000000ull, 0x1ef3622fba094800ull,
                                 0x1a22d9f300f5138full, 0xc63b05c06ca1493bull, 0x01ae3a4617c510eaull};
static const uint64_t P381[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};

static field_t g_fields[2];
static int g_init = 0;

static int fp_geq(const fp_t* a, const fp_t* b) {
  for (int i = NLIMB - 1; i >= 0; i--) {
    if (a->l[i] != b->l[i]) return a->l[i] > b->l[i];
  }
  return 1;
}

static uint64_t fp_add_raw(fp_t* r, const fp_t* a, const fp_t* b) {
  u128 c = 0;
  for (int i = 0; i < NLIMB; i++) {
    c += (u128)a->l[i] + b->l[i];
    r->l[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}

static uint64_t fp_sub_raw(fp_t* r, const fp_t* a, const fp_t* b) {
  uint64_t borrow = 0;
  for (int i = 0; i < NLIMB; i++) {
    u128 d = (u128)a->l[i] - b->l[i] - borrow;
    r->l[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}

/* montgomery_backend.rs:108-113 add_assign: add, then subtract the modulus if needed. */
static void fp_add(const field_t* f, fp_t* r, const fp_t* a, const fp_t* b) {
  fp_t t;
  uint64_t c = fp_add_raw(&t, a, b);
  if (c || fp_geq(&t, &f->p)) fp_sub_raw(&t, &t, &f->p);
  *r = t;
}

/* montgomery_backend.rs:115-129 sub_assign: add the modulus first when b > a. */
static void fp_sub(const field_t* f, fp_t* r, const fp_t* a, const fp_t* b) {
  fp_t t;
  if (fp_sub_raw(&t, a, b)) fp_add_raw(&t, &t, &f->p);
  *r = t;
}

static void fp_dbl(const field_t* f, fp_t* r, const fp_t* a) { fp_add(f, r, a, a); }

static void fp_neg(const field_t* f, fp_t* r, const fp_t* a) {
  fp_t z;
  memset(&z, 0, sizeof z);
  fp_sub(f, r, &z, a);
}

static int fp_is_zero(const fp_t* a) {
  uint64_t o = 0;
  for (int i = 0; i < NLIMB; i++) o |= a->l[i];
  return o == 0;
}

static int fp_eq(const fp_t* a, const fp_t* b) { return memcmp(a, b, sizeof *a) == 0; }

/* montgomery_backend.rs:146-201 mul_assign: CIOS over 64-bit limbs, then subtract_modulus. */
static void fp_mul(const field_t* f, fp_t* r, const fp_t* a, const fp_t* b) {
  uint64_t t[NLIMB + 2];
  memset(t, 0, sizeof t);
  for (int i = 0; i < NLIMB; i++) {
    u128 c = 0;
    for (int j = 0; j < NLIMB; j++) {
      c += (u128)a->l[j] * b->l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[NLIMB];
    t[NLIMB] = (uint64_t)c;
    t[NLIMB + 1] = (uint64_t)(c >> 64);
    uint64_t k = t[0] * f->inv;
    c = (u128)k * f->p.l[0] + t[0];
    c >>= 64;
    for (int j = 1; j < NLIMB; j++) {
      c += (u128)k * f->p.l[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[NLIMB];
    t[NLIMB - 1] = (uint64_t)c;
    t[NLIMB] = t[NLIMB + 1] + (uint64_t)(c >> 64);
  }
  fp_t out;
  memcpy(out.l, t, sizeof out.l);
  if (t[NLIMB] || fp_geq(&out, &f->p)) fp_sub_raw(&out, &out, &f->p);
  *r = out;
}

static void fp_sqr(const field_t* f, fp_t* r, const fp_t* a) { fp_mul(f, r, a, a); }

/* a^(p-2) (montgomery form in, montgomery form out). */
static void fp_inv(const field_t* f, fp_t* r, const fp_t* a) {
  fp_t e = f->p, two, acc = f->one;
  memset(&two, 0, sizeof two);
  two.l[0] = 2;
  fp_sub_raw(&e, &e, &two);
  for (int i = NLIMB * 64 - 1; i >= 0; i--) {
    fp_sqr(f, &acc, &acc);
    if ((e.l[i >> 6] >> (i & 63)) & 1) fp_mul(f, &acc, &acc, a);
  }
  *r = acc;
}

static void field_init(field_t* f, const uint64_t* p, int scalar_bits) {
  memcpy(f->p.l, p, sizeof f->p.l);
  f->scalar_bits = scalar_bits;
  uint64_t x = 1; /* Newton: x = p^-1 mod 2^64 */
  for (int i = 0; i < 6; i++) x *= 2 - p[0] * x;
  f->inv = (uint64_t)0 - x;
  /* one = 2^384 mod p by 384 modular doublings of 1; r2 by 384 more */
  fp_t v;
  memset(&v, 0, sizeof v);
  v.l[0] = 1;
  for (int i = 0; i < 384; i++) fp_dbl(f, &v, &v);
  f->one = v;
  for (int i = 0; i < 384; i++) fp_dbl(f, &v, &v);
  f->r2 = v;
}

static void oracle_init(void) {
  if (g_init) return;
  field_init(&g_fields[0], P377, 253);
  field_init(&g_fields[1], P381, 255);
  g_init = 1;
}

/* ---- Jacobian group law (a = 0) -------------------------------------------------------------- */
static void jac_zero(const field_t* f, jac_t* r) { /* short_weierstrass.rs:750-756: (1, 1, 0) */
  r->x = f->one;
  r->y = f->one;
  memset(&r->z, 0, sizeof r->z);
}

static int jac_is_zero(const jac_t* a) { return fp_is_zero(&a->z); }

/* short_weierstrass.rs:815-848 */
static void jac_double(const field_t* f, jac_t* s) {
  if (jac_is_zero(s)) return;
  fp_t a, b, c, d, e, ff, t;
  fp_sqr(f, &a, &s->x);         /* A = X1^2 */
  fp_sqr(f, &b, &s->y);         /* B = Y1^2 */
  fp_sqr(f, &c, &b);            /* C = B^2 */
  fp_add(f, &t, &s->x, &b);     /* D = 2*((X1+B)^2 - A - C) */
  fp_sqr(f, &t, &t);
  fp_sub(f, &t, &t, &a);
  fp_sub(f, &t, &t, &c);
  fp_dbl(f, &d, &t);
  fp_dbl(f, &t, &a);            /* E = 3*A */
  fp_add(f, &e, &a, &t);
  fp_sqr(f, &ff, &e);           /* F = E^2 */
  fp_mul(f, &s->z, &s->z, &s->y); /* Z3 = 2*Y1*Z1 */
  fp_dbl(f, &s->z, &s->z);
  fp_dbl(f, &t, &d);            /* X3 = F - 2*D */
  fp_sub(f, &s->x, &ff, &t);
  fp_sub(f, &t, &d, &s->x);     /* Y3 = E*(D - X3) - 8*C */
  fp_mul(f, &t, &t, &e);
  fp_dbl(f, &c, &c);
  fp_dbl(f, &c, &c);
  fp_dbl(f, &c, &c);
  fp_sub(f, &s->y, &t, &c);
}

/* short_weierstrass.rs:886-948 */
static void jac_add_mixed(const field_t* f, jac_t* s, const aff_t* o) {
  if (o->inf) return;
  if (jac_is_zero(s)) {
    s->x = o->x;
    s->y = o->y;
    s->z = f->one;
    return;
  }
  fp_t z1z1, u2, s2, h, hh, i, j, r, v, t, t2;
  fp_sqr(f, &z1z1, &s->z);
  fp_mul(f, &u2, &z1z1, &o->x);
  fp_mul(f, &s2, &s->z, &o->y);
  fp_mul(f, &s2, &s2, &z1z1);
  if (fp_eq(&s->x, &u2) && fp_eq(&s->y, &s2)) {
    jac_double(f, s);
    return;
  }
  fp_sub(f, &h, &u2, &s->x);    /* H = U2 - X1 */
  fp_sqr(f, &hh, &h);           /* HH = H^2 */
  fp_dbl(f, &i, &hh);           /* I = 4*HH */
  fp_dbl(f, &i, &i);
  fp_mul(f, &j, &h, &i);        /* J = H*I */
  fp_sub(f, &r, &s2, &s->y);    /* r = 2*(S2 - Y1) */
  fp_dbl(f, &r, &r);
  fp_mul(f, &v, &s->x, &i);     /* V = X1*I */
  fp_sqr(f, &t, &r);            /* X3 = r^2 - J - 2*V */
  fp_sub(f, &t, &t, &j);
  fp_dbl(f, &t2, &v);
  fp_sub(f, &t, &t, &t2);
  fp_t x3 = t;
  fp_sub(f, &t, &v, &x3);       /* Y3 = r*(V - X3) - 2*Y1*J */
  fp_mul(f, &t, &t, &r);
  fp_mul(f, &t2, &s->y, &j);
  fp_dbl(f, &t2, &t2);
  fp_sub(f, &s->y, &t, &t2);
  s->x = x3;
  fp_add(f, &t, &s->z, &h);     /* Z3 = (Z1 + H)^2 - Z1Z1 - HH */
  fp_sqr(f, &t, &t);
  fp_sub(f, &t, &t, &z1z1);
  fp_sub(f, &s->z, &t, &hh);
}

/* short_weierstrass.rs:979-1040 */
static void jac_add(const field_t* f, jac_t* s, const jac_t* o) {
  if (jac_is_zero(s)) {
    *s = *o;
    return;
  }
  if (jac_is_zero(o)) return;
  fp_t z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, t2;
  fp_sqr(f, &z1z1, &s->z);
  fp_sqr(f, &z2z2, &o->z);
  fp_mul(f, &u1, &s->x, &z2z2);
  fp_mul(f, &u2, &o->x, &z1z1);
  fp_mul(f, &s1, &s->y, &o->z);
  fp_mul(f, &s1, &s1, &z2z2);
  fp_mul(f, &s2, &o->y, &s->z);
  fp_mul(f, &s2, &s2, &z1z1);
  if (fp_eq(&u1, &u2) && fp_eq(&s1, &s2)) {
    jac_double(f, s);
    return;
  }
  fp_sub(f, &h, &u2, &u1);      /* H = U2 - U1 */
  fp_dbl(f, &i, &h);            /* I = (2H)^2 */
  fp_sqr(f, &i, &i);
  fp_mul(f, &j, &h, &i);        /* J = H*I */
  fp_sub(f, &r, &s2, &s1);      /* r = 2*(S2 - S1) */
  fp_dbl(f, &r, &r);
  fp_mul(f, &v, &u1, &i);       /* V = U1*I */
  fp_sqr(f, &t, &r);            /* X3 = r^2 - J - 2V */
  fp_sub(f, &t, &t, &j);
  fp_dbl(f, &t2, &v);
  fp_sub(f, &t, &t, &t2);
  fp_t x3 = t;
  fp_sub(f, &t, &v, &x3);       /* Y3 = r*(V - X3) - 2*S1*J */
  fp_mul(f, &t, &t, &r);
  fp_mul(f, &t2, &s1, &j);
  fp_dbl(f, &t2, &t2);
  fp_t y3;
  fp_sub(f, &y3, &t, &t2);
  fp_add(f, &t, &s->z, &o->z);  /* Z3 = ((Z1 + Z2)^2 - Z1Z1 - Z2Z2)*H */
  fp_sqr(f, &t, &t);
  fp_sub(f, &t, &t, &z1z1);
  fp_sub(f, &t, &t, &z2z2);
  fp_mul(f, &s->z, &t, &h);
  s->x = x3;
  s->y = y3;
}

/* short_weierstrass.rs:1093-1115, written back as a normalised Projective image:
 * (x, y, 1) or the zero() triple (1, 1, 0). */
static void jac_write_normalized(const field_t* f, const jac_t* a, uint8_t* out144) {
  jac_t r;
  if (jac_is_zero(a)) {
    jac_zero(f, &r);
  } else {
    fp_t zi, zi2, zi3;
    fp_inv(f, &zi, &a->z);
    fp_sqr(f, &zi2, &zi);
    fp_mul(f, &zi3, &zi2, &zi);
    fp_mul(f, &r.x, &a->x, &zi2);
    fp_mul(f, &r.y, &a->y, &zi3);
    r.z = f->one;
  }
  memcpy(out144, r.x.l, 48);
  memcpy(out144 + 48, r.y.l, 48);
  memcpy(out144 + 96, r.z.l, 48);
}

static void aff_read(aff_t* a, const uint8_t* p) {
  memcpy(a->x.l, p, 48);
  memcpy(a->y.l, p + 48, 48);
  a->inf = p[96] != 0; /* the flag byte is authoritative (short_weierstrass.rs:127-135) */
}

/* ---- msm_bigint (variable_base/mod.rs:68-162) -------------------------------------------------- */
static int ark_log2_ceil(size_t n) { /* ark_std::log2: ceil(log2 n), 0 for n <= 1 */
  int lg = 0;
  while (((size_t)1 << lg) < n) lg++;
  return lg;
}

int oracle_window_bits(size_t size) { /* mod.rs:77-81 + msm/mod.rs:54-57 */
  if (size < 32) return 3;
  return ark_log2_ceil(size) * 69 / 100 + 2;
}

typedef struct {
  const field_t* f;
  const uint8_t* bases;
  size_t stride;
  const uint8_t* scalars;
  size_t n;
  int c;
  int w_start;
  jac_t result;
} win_job_t;

static int scalar_is_zero(const uint64_t* k) { return (k[0] | k[1] | k[2] | k[3]) == 0; }
static int scalar_is_one(const uint64_t* k) { return k[0] == 1 && (k[1] | k[2] | k[3]) == 0; }

/* (scalar >> w_start) limb 0, then % 2^c  (mod.rs:107-113: divn then as_ref()[0] % (1 << c)) */
static uint64_t scalar_window(const uint64_t* k, int w_start, int c) {
  int limb = w_start >> 6, sh = w_start & 63;
  uint64_t v = limb < 4 ? k[limb] >> sh : 0;
  if (sh && limb + 1 < 4) v |= k[limb + 1] << (64 - sh);
  return v & (((uint64_t)1 << c) - 1);
}

static void* window_job(void* arg) {
  win_job_t* j = (win_job_t*)arg;
  const field_t* f = j->f;
  size_t nb = ((size_t)1 << j->c) - 1;
  jac_t* buckets = (jac_t*)malloc(nb * sizeof(jac_t));
  for (size_t b = 0; b < nb; b++) jac_zero(f, &buckets[b]);
  jac_t res;
  jac_zero(f, &res);
  for (size_t i = 0; i < j->n; i++) {
    uint64_t k[4];
    memcpy(k, j->scalars + 32 * i, 32);
    if (scalar_is_zero(k)) continue; /* mod.rs:75 */
    aff_t base;
    aff_read(&base, j->bases + i * j->stride);
    if (scalar_is_one(k)) { /* mod.rs:100-104 */
      if (j->w_start == 0) jac_add_mixed(f, &res, &base);
      continue;
    }
    uint64_t d = scalar_window(k, j->w_start, j->c);
    if (d != 0) jac_add_mixed(f, &buckets[d - 1], &base); /* mod.rs:118-120 */
  }
  jac_t running;
  jac_zero(f, &running);
  for (size_t b = nb; b-- > 0;) { /* mod.rs:138-142 */
    jac_add(f, &running, &buckets[b]);
    jac_add(f, &res, &running);
  }
  free(buckets);
  j->result = res;
  return NULL;
}

/* Returns 0 on success.  threads <= 0 means one thread per window. */
int oracle_msm(int curve, const uint8_t* bases, size_t stride, const uint8_t* scalars, size_t n, uint8_t* out144,
               int threads) {
  if (curve < 0 || curve > 1) return -1;
  oracle_init();
  const field_t* f = &g_fields[curve];
  int c = oracle_window_bits(n);
  int nwin = (f->scalar_bits + c - 1) / c; /* (0..num_bits).step_by(c), mod.rs:87 */
  win_job_t* jobs = (win_job_t*)calloc(nwin, sizeof(win_job_t));
  pthread_t* tids = (pthread_t*)calloc(nwin, sizeof(pthread_t));
  if (threads <= 0 || threads > nwin) threads = nwin;
  for (int w = 0; w < nwin; w++) {
    jobs[w].f = f;
    jobs[w].bases = bases;
    jobs[w].stride = stride;
    jobs[w].scalars = scalars;
    jobs[w].n = n;
    jobs[w].c = c;
    jobs[w].w_start = w * c;
  }
  for (int w0 = 0; w0 < nwin; w0 += threads) {
    int w1 = w0 + threads < nwin ? w0 + threads : nwin;
    if (threads == 1) {
      window_job(&jobs[w0]);
      continue;
    }
    for (int w = w0; w < w1; w++) pthread_create(&tids[w], NULL, window_job, &jobs[w]);
    for (int w = w0; w < w1; w++) pthread_join(tids[w], NULL);
  }
  /* mod.rs:148-161: lowest + fold(rev windows[1..]) */
  jac_t total;
  jac_zero(f, &total);
  for (int w = nwin - 1; w >= 1; w--) {
    jac_add(f, &total, &jobs[w].result);
    for (int i = 0; i < c; i++) jac_double(f, &total);
  }
  jac_t lowest = jobs[0].result;
  jac_add(f, &lowest, &total);
  jac_write_normalized(f, &lowest, out144);
  free(jobs);
  free(tids);
  return 0;
}

/* sum k_i P_i by plain double-and-add: the property ARK test-templates/src/msm.rs:7-38 checks msm against. */
int oracle_msm_naive(int curve, const uint8_t* bases, size_t stride, const uint8_t* scalars, size_t n, uint8_t* out144) {
  if (curve < 0 || curve > 1) return -1;
  oracle_init();
  const field_t* f = &g_fields[curve];
  jac_t total;
  jac_zero(f, &total);
  for (size_t i = 0; i < n; i++) {
    uint64_t k[4];
    memcpy(k, scalars + 32 * i, 32);
    aff_t base;
    aff_read(&base, bases + i * stride);
    jac_t r;
    jac_zero(f, &r);
    for (int bit = 255; bit >= 0; bit--) {
      jac_double(f, &r);
      if ((k[bit >> 6] >> (bit & 63)) & 1) jac_add_mixed(f, &r, &base);
    }
    jac_add(f, &total, &r);
  }
  jac_write_normalized(f, &total, out144);
  return 0;
}

/* Field and group primitives exported so tests can pin them against the reference's constants and oracle/_ref. */
int oracle_fp_mul(int curve, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  if (curve < 0 || curve > 1) return -1;
  oracle_init();
  fp_t x, y, z;
  memcpy(x.l, a, 48);
  memcpy(y.l, b, 48);
  fp_mul(&g_fields[curve], &z, &x, &y);
  memcpy(out, z.l, 48);
  return 0;
}

/* consts: p (48) | R mod p (48) | R^2 mod p (48) | inv (8) */
int oracle_field_consts(int curve, uint8_t* out152) {
  if (curve < 0 || curve > 1) return -1;
  oracle_init();
  const field_t* f = &g_fields[curve];
  memcpy(out152, f->p.l, 48);
  memcpy(out152 + 48, f->one.l, 48);
  memcpy(out152 + 96, f->r2.l, 48);
  memcpy(out152 + 144, &f->inv, 8);
  return 0;
}

/* out = a + b for two Affine images (through add_assign_mixed on a Jacobian copy of a). */
int oracle_affine_add(int curve, const uint8_t* a104, const uint8_t* b104, uint8_t* out144) {
  if (curve < 0 || curve > 1) return -1;
  oracle_init();
  const field_t* f = &g_fields[curve];
  aff_t a, b;
  aff_read(&a, a104);
  aff_read(&b, b104);
  jac_t s;
  jac_zero(f, &s);
  jac_add_mixed(f, &s, &a);
  jac_add_mixed(f, &s, &b);
  jac_write_normalized(f, &s, out144);
  return 0;
}

/* Generate n subgroup points P_i = (h0 + i*h1) * G as Affine images (stride 104), for synthetic benches:
 * "distinct" points are produced once and the vector is replicated by doubling, the shape of the reference
 * generator (P1A yrrid/src/util.rs:15-28). */
int oracle_gen_points(int curve, const uint8_t* gen104, const uint8_t* h0_32, const uint8_t* h1_32, size_t distinct,
                      size_t n, uint8_t* out) {
  if (curve < 0 || curve > 1) return -1;
  oracle_init();
  const field_t* f = &g_fields[curve];
  aff_t g;
  aff_read(&g, gen104);
  jac_t acc, step;
  const uint8_t* hs[2] = {h0_32, h1_32};
  jac_t* outs[2] = {&acc, &step};
  for (int s = 0; s < 2; s++) {
    uint64_t k[4];
    memcpy(k, hs[s], 32);
    jac_t r;
    jac_zero(f, &r);
    for (int bit = 255; bit >= 0; bit--) {
      jac_double(f, &r);
      if ((k[bit >> 6] >> (bit & 63)) & 1) jac_add_mixed(f, &r, &g);
    }
    *outs[s] = r;
  }
  if (distinct > n) distinct = n;
  for (size_t i = 0; i < distinct; i++) {
    uint8_t img[144];
    jac_write_normalized(f, &acc, img);
    uint8_t* o = out + 104 * i;
    memset(o, 0, 104);
    if (jac_is_zero(&acc)) {
      o[96] = 1;
    } else {
      memcpy(o, img, 96);
    }
    jac_add(f, &acc, &step);
  }
  for (size_t have = distinct; have < n;) {
    size_t cp = have < n - have ? have : n - have;
    memcpy(out + 104 * have, out, 104 * cp);
    have += cp;
  }
  return 0;
}
