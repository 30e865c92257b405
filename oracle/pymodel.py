"""Python big-integer model of the MSM path.  TEST INFRASTRUCTURE ONLY.

This file is part of ``oracle/``: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker.  It is
deliberately *independent* of the C restatement in ``msm_oracle.c``: it works on
affine points with plain chord-and-tangent arithmetic over Python ints, so an error
in the Jacobian/Montgomery restatement cannot hide behind a matching error here.

What it models (reference file:line, all under /root/reference):
  * curve constants          ARKC bls12_377/src/fields/fq.rs:4, fr.rs:24, curves/g1.rs:28-42,153-159,
                             bls12_381/src/fields/fq.rs:4, fr.rs:4, curves/g1.rs:37,69-73,
                             bls12_377/src/fields/fq2.rs:13, curves/g2.rs:47-78,
                             bls12_381/src/fields/fq2.rs:13, curves/g2.rs:47-48,74-91
  * Montgomery constants     SPK ff/bls12-377.hpp:10-25, ff/bls12-381.hpp:10-25 (R = 2^384, M0)
  * the MSM contract         ARK ec/src/msm/variable_base/mod.rs:68-162 (result = sum k_i * P_i,
                             truncated to the shorter slice, zero scalars skipped)
  * ABI layouts              arkworks G1Affine 104 B (x, y Montgomery 6xu64 LE, infinity byte, 7 pad),
                             BigInteger256 32 B LE, G1Projective 144 B (x, y, z Montgomery);
                             SURVEY.md section 8(b).
"""
from __future__ import annotations

import random
import struct
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

R_BITS = 384
R = 1 << R_BITS


# ----------------------------------------------------------------------------- fields
class Fp2:
    """Element c0 + c1*u of Fp[u]/(u^2 - nonresidue)."""
    __slots__ = ("c0", "c1", "p", "nr")

    def __init__(self, c0, c1, p, nr):
        self.c0, self.c1, self.p, self.nr = c0 % p, c1 % p, p, nr

    def _w(self, a, b):
        return Fp2(a, b, self.p, self.nr)

    def __add__(self, o):
        return self._w(self.c0 + o.c0, self.c1 + o.c1)

    def __sub__(self, o):
        return self._w(self.c0 - o.c0, self.c1 - o.c1)

    def __neg__(self):
        return self._w(-self.c0, -self.c1)

    def __mul__(self, o):
        if isinstance(o, int):
            return self._w(self.c0 * o, self.c1 * o)
        return self._w(self.c0 * o.c0 + self.nr * self.c1 * o.c1, self.c0 * o.c1 + self.c1 * o.c0)

    def __eq__(self, o):
        return self.c0 == o.c0 and self.c1 == o.c1

    def is_zero(self):
        return self.c0 == 0 and self.c1 == 0

    def inv(self):
        n = (self.c0 * self.c0 - self.nr * self.c1 * self.c1) % self.p
        ni = pow(n, -1, self.p)
        return self._w(self.c0 * ni, -self.c1 * ni)


@dataclass(frozen=True)
class Curve:
    name: str
    p: int          # base-field modulus
    r: int          # scalar-field modulus (group order of the prime subgroup)
    b: object       # int (G1) or (c0, c1) tuple (G2)
    gx: object
    gy: object
    ext: int = 1    # extension degree of the coordinate field (1 = G1, 2 = G2)
    nonresidue: int = 0
    curve_id: int = 0

    # -- coordinate-field helpers -------------------------------------------------
    def F(self, v):
        if self.ext == 1:
            return v % self.p
        return Fp2(v[0], v[1], self.p, self.nonresidue % self.p)

    def f_zero(self):
        return self.F(0 if self.ext == 1 else (0, 0))

    def f_is_zero(self, v):
        return v == 0 if self.ext == 1 else v.is_zero()

    def f_inv(self, v):
        return pow(v, -1, self.p) if self.ext == 1 else v.inv()

    def f_mul(self, a, b):
        return (a * b) % self.p if self.ext == 1 else a * b

    def f_add(self, a, b):
        return (a + b) % self.p if self.ext == 1 else a + b

    def f_sub(self, a, b):
        return (a - b) % self.p if self.ext == 1 else a - b

    def f_neg(self, a):
        return (-a) % self.p if self.ext == 1 else -a

    def f_small(self, a, k):
        return (a * k) % self.p if self.ext == 1 else a * k

    @property
    def scalar_bits(self):
        return self.r.bit_length()

    @property
    def coord_bytes(self):
        return 48 * self.ext

    @property
    def affine_stride(self):          # size_of::<Affine>() in arkworks: 2 coords + flag, padded to 8
        return 2 * self.coord_bytes + 8

    @property
    def projective_bytes(self):
        return 3 * self.coord_bytes

    def generator(self):
        return (self.F(self.gx), self.F(self.gy))

    def on_curve(self, P):
        if P is None:
            return True
        x, y = P
        lhs = self.f_mul(y, y)
        rhs = self.f_add(self.f_mul(self.f_mul(x, x), x), self.F(self.b))
        return lhs == rhs

    # -- group law (affine, None = infinity) ---------------------------------------
    def neg(self, P):
        return None if P is None else (P[0], self.f_neg(P[1]))

    def add(self, P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if y1 == y2 and not self.f_is_zero(y1):
                lam = self.f_mul(self.f_small(self.f_mul(x1, x1), 3), self.f_inv(self.f_small(y1, 2)))
            else:
                return None          # P == -Q, or a 2-torsion point doubled
        else:
            lam = self.f_mul(self.f_sub(y2, y1), self.f_inv(self.f_sub(x2, x1)))
        x3 = self.f_sub(self.f_sub(self.f_mul(lam, lam), x1), x2)
        y3 = self.f_sub(self.f_mul(lam, self.f_sub(x1, x3)), y1)
        return (x3, y3)

    def mul(self, k, P):
        Q = None
        while k:
            if k & 1:
                Q = self.add(Q, P)
            P = self.add(P, P)
            k >>= 1
        return Q

    def msm_naive(self, bases, scalars):
        """sum k_i*P_i by double-and-add; the property ARK test-templates/src/msm.rs:7-38 checks."""
        n = min(len(bases), len(scalars))
        acc = None
        for i in range(n):
            if scalars[i]:
                acc = self.add(acc, self.mul(scalars[i], bases[i]))
        return acc

    def msm_pippenger(self, bases, scalars, c=None):
        """Windowed bucket method with arkworks' window rule (variable_base/mod.rs:77-161), on affine ints."""
        n = min(len(bases), len(scalars))
        if c is None:
            c = 3 if n < 32 else ark_window_bits(n)
        sums = []
        for w_start in range(0, self.scalar_bits, c):
            buckets = [None] * ((1 << c) - 1)
            res = None
            for i in range(n):
                k = scalars[i]
                if k == 0:
                    continue
                if k == 1:
                    if w_start == 0:
                        res = self.add(res, bases[i])
                    continue
                d = ((k >> w_start) & ((1 << 64) - 1)) % (1 << c)
                if d:
                    buckets[d - 1] = self.add(buckets[d - 1], bases[i])
            run = None
            for bkt in reversed(buckets):
                run = self.add(run, bkt)
                res = self.add(res, run)
            sums.append(res)
        total = None
        for s in reversed(sums[1:]):
            total = self.add(total, s)
            for _ in range(c):
                total = self.add(total, total)
        return self.add(sums[0], total)

    # -- ABI encodings ---------------------------------------------------------------
    def _enc_f(self, v) -> bytes:
        if self.ext == 1:
            return ((v * R) % self.p).to_bytes(48, "little")
        return ((v.c0 * R) % self.p).to_bytes(48, "little") + ((v.c1 * R) % self.p).to_bytes(48, "little")

    def _dec_f(self, b: bytes):
        rinv = pow(R, -1, self.p)
        if self.ext == 1:
            return (int.from_bytes(b[:48], "little") * rinv) % self.p
        return self.F(((int.from_bytes(b[:48], "little") * rinv) % self.p,
                       (int.from_bytes(b[48:96], "little") * rinv) % self.p))

    def encode_affine(self, P, zero_style="0.4") -> bytes:
        """arkworks Affine image: x, y in Montgomery form, infinity flag byte, 7 pad bytes."""
        cb = self.coord_bytes
        if P is None:
            # 0.4-dev zero is (0,0,true) (ARK short_weierstrass.rs:185-191); 0.3 is (0,1,true).
            y = self._enc_f(self.F(1 if self.ext == 1 else (1, 0))) if zero_style == "0.3" else bytes(cb)
            return bytes(cb) + y + b"\x01" + bytes(7)
        return self._enc_f(P[0]) + self._enc_f(P[1]) + b"\x00" + bytes(7)

    def decode_affine(self, b: bytes):
        cb = self.coord_bytes
        if b[2 * cb] != 0:
            return None
        return (self._dec_f(b[:cb]), self._dec_f(b[cb:2 * cb]))

    def encode_affine_array(self, pts, zero_style="0.4") -> bytes:
        return b"".join(self.encode_affine(P, zero_style) for P in pts)

    def encode_projective_normalized(self, P) -> bytes:
        """Canonical result image: (X, Y, Z=1) in Montgomery form; infinity is (1,1,0) (ARK :750-756)."""
        one = self.F(1 if self.ext == 1 else (1, 0))
        if P is None:
            return self._enc_f(one) + self._enc_f(one) + bytes(self.coord_bytes)
        return self._enc_f(P[0]) + self._enc_f(P[1]) + self._enc_f(one)

    def encode_serialized(self, P) -> bytes:
        """arkworks CanonicalSerialize uncompressed record: x | y little-endian normal form, SWFlags in the last byte's
        top bits (infinity = 0x40; P1B nickray driver/algebra/serialize/src/flags.rs:107-134)."""
        cb = self.coord_bytes
        if P is None:
            # GroupAffine::zero() = (0, 1, infinity = true) in the 0.3-era tree the harness is built on: x = 0, y = 1 (normal
            # form; Fq2: y.c0 = 1), flag ORed into the last byte (P1B nickray .../short_weierstrass_jacobian.rs:154-156, 827-835)
            rec = bytearray(2 * cb)
            rec[cb] = 1
            rec[-1] |= 0x40
            return bytes(rec)
        def enc(v):
            if self.ext == 1:
                return v.to_bytes(48, "little")
            return v.c0.to_bytes(48, "little") + v.c1.to_bytes(48, "little")
        return enc(P[0]) + enc(P[1])

    def decode_projective(self, b: bytes):
        """Jacobian (X, Y, Z) image -> affine model point (x = X/Z^2, y = Y/Z^3; ARK :1093-1115)."""
        cb = self.coord_bytes
        X, Y, Z = self._dec_f(b[:cb]), self._dec_f(b[cb:2 * cb]), self._dec_f(b[2 * cb:3 * cb])
        if self.f_is_zero(Z):
            return None
        zi = self.f_inv(Z)
        zi2 = self.f_mul(zi, zi)
        return (self.f_mul(X, zi2), self.f_mul(Y, self.f_mul(zi2, zi)))


def ark_window_bits(n: int) -> int:
    """ln_without_floats(n) + 2 with ark_std::log2 = ceil(log2) (ARK ec/src/msm/mod.rs:54-57)."""
    lg = (n - 1).bit_length() if n > 1 else 0
    return lg * 69 // 100 + 2


def encode_scalars(scalars: Sequence[int]) -> bytes:
    return b"".join(int(k).to_bytes(32, "little") for k in scalars)


def decode_scalars(b: bytes) -> List[int]:
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


BLS12_377_G1 = Curve(
    name="bls12_377_g1",
    p=258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177,
    r=8444461749428370424248824938781546531375899335154063827935233455917409239041,
    b=1,
    gx=81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
    gy=241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030,
    curve_id=0,
)

BLS12_381_G1 = Curve(
    name="bls12_381_g1",
    p=4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559787,
    r=52435875175126190479447740508185965837690552500527637822603658699938581184513,
    b=4,
    gx=3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507,
    gy=1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569,
    curve_id=1,
)

BLS12_377_G2 = Curve(
    name="bls12_377_g2",
    p=BLS12_377_G1.p,
    r=BLS12_377_G1.r,
    b=(0, 155198655607781456406391640216936120121836107652948796323930557600032281009004493664981332883744016074664192874906),
    gx=(233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
        140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118),
    gy=(63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
        149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491),
    ext=2,
    nonresidue=-5,
    curve_id=2,
)

# ARKC bls12_381/src/curves/g2.rs:47-48 (COEFF_B = (4, 4)), :74-91 (generator), fields/fq2.rs:13 (NONRESIDUE = -1)
BLS12_381_G2 = Curve(
    name="bls12_381_g2",
    p=BLS12_381_G1.p,
    r=BLS12_381_G1.r,
    b=(4, 4),
    gx=(352701069587466618187139116011060144890029952792775240219908644239793785735715026873347600343865175952761926303160,
        3059144344244213709971259814753781636986470325476647558659373206291635324768958432433509563104347017837885763365758),
    gy=(1985150602287291935568054521177171638300868978215655730859378665066344726373823718423869104263333984641494340347905,
        927553665492332455747201965776037880757740193453592970025027978793976877002675564980949289727957565575433344219582),
    ext=2,
    nonresidue=-1,
    curve_id=3,
)

CURVES = {c.name: c for c in (BLS12_377_G1, BLS12_381_G1, BLS12_377_G2, BLS12_381_G2)}


def blst377_ring_curve(seed: int = 1) -> Curve:
    """NOT a curve of the product (curve id 4 of msm_oracle.c, tests only): y^2 = x^3 + b over the RING Fp[u]/(u^2 + 1) with the
    BLS12-377 prime -- the coordinate structure the blst copy under /root/reference computes its "G2" in (its author changed the
    modulus to BLS12-377's and left the tower at u^2 = -1; oracle/ref_driver_blst377.c).  p = 1 mod 4, so the ring is Fp x Fp and
    the "curve" a pair of curves over Fp; the group law holds componentwise.  The base point is a seeded random pair (x, y) and
    b := y^2 - x^3 (the a = 0 formulas never use b); `r` only bounds the synthetic scalars."""
    rng = random.Random(seed)
    p = BLS12_377_G1.p
    x = (rng.randrange(p), rng.randrange(p))
    y = (rng.randrange(p), rng.randrange(p))
    fx, fy = Fp2(x[0], x[1], p, -1 % p), Fp2(y[0], y[1], p, -1 % p)
    b = fy * fy - fx * fx * fx
    return Curve(name="blst377_ring_g2", p=p, r=BLS12_377_G1.r, b=(b.c0, b.c1), gx=x, gy=y, ext=2, nonresidue=-1, curve_id=4)
CURVES_BY_ID = {c.curve_id: c for c in CURVES.values()}

# Literal points of the FPGA harness edge-case tests (hex, normal form, BLS12-377 G1):
# P1B hardcaml/zprize/msm_pippenger/test_fpga_harness/tests/msm_unit_tests.rs:30-49
EDGE_P = (int("32D756062D349E59416ECE15CCBF8E86EF0D33183465A42FE2CB65FC1664272E6BB28F0E1C7A7C9C05824AD09ADC00", 16),
          int("6E4B66BB23EF4BEF715F597162D6662D8161CD062D6212D39392E17232444A0760B5DC479DB98123AB3887AA3CB34E", 16))
EDGE_P_NEG = (int("32d756062d349e59416ece15ccbf8e86ef0d33183465a42fe2cb65fc1664272e6bb28f0e1c7a7c9c05824ad09adc00", 16),
              int("13feedf5ca1219ed6c9a666fb3e72d4eca17825fac7b17c4b5fcf4e47d703b60faaa767e862467f615d877855c34cb3", 16))
EDGE_T = (int("1ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000000", 16), 0)


# ----------------------------------------------------------------------------- synthetic inputs
def random_scalars(curve: Curve, n: int, rng: random.Random) -> List[int]:
    """Uniform integers in [0, r): what the harness feeds after transmuting Fr to BigInteger256."""
    return [rng.randrange(curve.r) for _ in range(n)]


def random_points(curve: Curve, n: int, rng: random.Random, distinct: Optional[int] = None):
    """n subgroup points; like the reference generator (P1A yrrid/src/util.rs:15-28) a small set of
    distinct points is replicated by doubling the vector, so equal points meet in buckets."""
    distinct = min(n, distinct or n)
    g = curve.generator()
    base = []
    acc = curve.mul(rng.randrange(1, curve.r), g)
    step = curve.mul(rng.randrange(1, curve.r), g)
    for _ in range(distinct):
        base.append(acc)
        acc = curve.add(acc, step)
    out = list(base)
    while len(out) < n:
        out.extend(out[: n - len(out)])
    return out
