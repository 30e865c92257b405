#!/usr/bin/env python3
"""Generate tests/golden/*.json: seeded MSM input/output vectors and reference-held constants.

Run in THIS container (it cross-checks every vector against oracle/_ref, the reference's own C/C++ sources built
from /root/reference, before writing):   python tools/gen_golden.py

Vectors are data only: inputs as the byte images the FFI passes (hex), expected output as the normalised 144-byte
projective image (hex).  Expected values come from oracle/pymodel.py (affine big-int arithmetic) and are asserted
equal to (a) the reference HostCurve naive MSM AND the reference's blst copy (its Pippenger, re-targeted by its author to the
BLS12-377 prime: oracle/ref_driver_blst377.c) for BLS12-377 G1, (b) the reference yrrid C MSM for BLS12-381 G1.  G2 vectors are
asserted equal to the C oracle, whose Fp2 template instance is itself checked against the reference's compiled G2 Pippenger
(tests/test_oracle.py) and, for BLS12-381, against the RFC 9380 vectors the reference holds (tests/golden/h2c_kat_bls12_381.json).

msm_vectors_large.json (sizes 2^10 and 2^12, SURVEY 8c "sizes 2^4 ... 2^12") keeps the FFI images compact: the `distinct` base
records once (the test replicates them by doubling the vector, exactly what the reference generator does: P1A
yrrid/src/util.rs:15-28) and every scalar explicitly.
"""
import ctypes
import json
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pymodel as m  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def ref377(curve, pts, sc):
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref377.so"))
    lib.ref377_msm_naive.restype = ctypes.c_int
    out = ctypes.create_string_buffer(144)
    inf = lib.ref377_msm_naive(curve.encode_affine_array(pts), ctypes.c_size_t(104), m.encode_scalars(sc),
                               ctypes.c_size_t(len(pts)), out)
    return curve.encode_projective_normalized(None) if inf else out.raw


def blst377(curve, pts, sc):
    """BLS12-377 G1 through the reference's blst Pippenger (raw Jacobian out, normalised by the C oracle's to-affine)."""
    sz = ctypes.c_size_t
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblst377.so"))
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    raw, out = ctypes.create_string_buffer(144), ctypes.create_string_buffer(144)
    lib.refblst_g1_msm(curve.encode_affine_array(pts), sz(104), m.encode_scalars(sc), sz(len(pts)), sz(253), raw)
    assert orc.oracle_jac_normalize(0, raw, out) == 0
    return out.raw


def oracle_c(curve, bases: bytes, scalars: bytes, n):
    sz = ctypes.c_size_t
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    out = ctypes.create_string_buffer(curve.projective_bytes)
    assert orc.oracle_msm(curve.curve_id, bases, sz(curve.affine_stride), scalars, sz(n), out, 0) == 0
    return out.raw


def ref381(curve, pts, sc):
    """Run the reference's C MSM binary; it prints the affine result in normal form."""
    if any(p is None for p in pts):
        return None
    exe = os.path.join(ROOT, "oracle", "_ref", "yrrid381_msm")
    with tempfile.TemporaryDirectory() as d:
        os.mkdir(os.path.join(d, "data"))
        with open(os.path.join(d, "data", "points.hex"), "w") as f:
            for P in pts:
                f.write("%x\n%x\n" % (P[0], P[1]))
        with open(os.path.join(d, "data", "scalars.hex"), "w") as f:
            for k in sc:
                f.write("%x\n" % k)
        r = subprocess.run([exe, str(len(pts))], cwd=d, capture_output=True, text=True, check=True)
    xs = [ln.split("=")[1].strip() for ln in r.stdout.splitlines() if ln.strip().startswith("x=")]
    ys = [ln.split("=")[1].strip() for ln in r.stdout.splitlines() if ln.strip().startswith("y=")]
    assert xs[0] == xs[1] and ys[0] == ys[1]
    return curve.encode_projective_normalized((int(xs[0], 16), int(ys[0], 16)))


def case(name, curve, pts, sc, note, zero_style="0.4", check_ref=True):
    exp = curve.encode_projective_normalized(curve.msm_naive(pts, sc))
    if len(pts) >= 32:
        assert exp == curve.encode_projective_normalized(curve.msm_pippenger(pts, sc))
    if check_ref:
        if curve.curve_id == 0:
            assert ref377(curve, pts, sc) == exp, name
            if all(k < (1 << 253) for k in sc):
                assert blst377(curve, pts, sc) == exp, name
        elif all(k < (1 << 255) for k in sc) and exp != curve.encode_projective_normalized(None):
            # (the reference binary has no printable form for an infinite result)
            r = ref381(curve, pts, sc)
            assert r is None or r == exp, name
    assert oracle_c(curve, curve.encode_affine_array(pts, zero_style), m.encode_scalars(sc), len(pts)) == exp, name
    return {"name": name, "curve": curve.name, "note": note, "n": len(pts),
            "bases": curve.encode_affine_array(pts, zero_style).hex(), "scalars": m.encode_scalars(sc).hex(),
            "expected": exp.hex()}


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = []
    for curve in (m.BLS12_377_G1, m.BLS12_381_G1):
        rng = random.Random(0xC0FFEE + curve.curve_id)
        for n, distinct in ((1, 1), (2, 2), (7, 3), (31, 31), (32, 8), (100, 10), (257, 64)):
            pts = m.random_points(curve, n, rng, distinct)
            sc = m.random_scalars(curve, n, rng)
            cases.append(case(f"random_n{n}", curve, pts, sc, "uniform scalars < r; replicated bases"))
        pts = m.random_points(curve, 40, rng, 5)
        sc = m.random_scalars(curve, 40, rng)
        sc[0], sc[1], sc[2], sc[3] = 0, 1, curve.r - 1, 2
        cases.append(case("special_scalars", curve, pts, sc, "0, 1, r-1, 2 among random scalars"))
        pts2 = list(pts)
        pts2[4] = None
        pts2[17] = None
        cases.append(case("with_infinity_ark04", curve, pts2, sc, "infinity bases, ark 0.4 zero (0,0,flag)"))
        cases.append(case("with_infinity_ark03", curve, pts2, sc, "infinity bases, ark 0.3 zero (0,1,flag)", zero_style="0.3"))
        cases.append(case("all_same_scalar", curve, pts, [sc[5]] * 40, "one hot bucket per window"))
        cases.append(case("all_zero_scalars", curve, pts, [0] * 40, "result is infinity"))
        g = curve.generator()
        alt = [g if i % 2 == 0 else curve.neg(g) for i in range(64)]
        cases.append(case("alternating_pm_generator", curve, alt, [sc[6]] * 64,
                          "P1B msm_unit_tests.rs:139-183: +G, -G with one scalar -> infinity"))
        cases.append(case("top_bits", curve, pts[:8], [(1 << (curve.scalar_bits - 1)) + i for i in range(8)], "top scalar bit set"))
    c = m.BLS12_377_G1
    rng = random.Random(99)
    rmont = (1 << 256) % c.r  # Fr::new(1) transmuted to BigInteger256 = R mod r
    cases.append(case("fpga_edge1", c, [m.EDGE_P, m.EDGE_P_NEG, m.EDGE_T, m.EDGE_T], [rmont] * 4,
                      "P1B msm_unit_tests.rs:21-77: P, -P, 2-torsion T twice -> infinity"))
    q = c.neg(c.add(m.EDGE_P, m.EDGE_T))
    cases.append(case("fpga_edge2", c, [m.EDGE_P, m.EDGE_T, q, m.EDGE_T], [rmont] * 4,
                      "P1B msm_unit_tests.rs:83-133: three points summing to infinity plus T"))
    cases.append(case("fpga_edge5", c, [c.generator(), m.EDGE_P], [1, 2], "P1B msm_unit_tests.rs:246+: G + 2*P"))
    g = c.generator()
    cases.append(case("trivial_inputs", c, [g] * 64, [1] + [0] * 63,
                      "P1B test_fpga_harness/src/util.rs:78-98 TEST_TRIVIAL_INPUTS: all bases = G, scalar 1 then zeros -> G"))
    assert cases[-1]["expected"] == c.encode_projective_normalized(g).hex()
    pts = m.random_points(c, 64, rng, 16)
    pts[16], pts[31], pts[47], pts[63] = m.EDGE_P, m.EDGE_T, m.EDGE_P_NEG, m.EDGE_T
    cases.append(case("fpga_edge4_boundaries", c, pts, m.random_scalars(c, 64, rng),
                      "P1B msm_unit_tests.rs:187-242 (scaled down): special points at chunk boundaries"))
    # BLS12-377 G2 (Fq2 coordinates, 200-byte Affine images); pinned by the Python model and the C oracle, whose Fp2
    # arithmetic is pinned by the reference's Fq2 KATs (tests/golden/fq2_kat_bls12_381.json)
    g2 = m.BLS12_377_G2
    rng = random.Random(0xC0FFEE + 2)
    for n, distinct in ((1, 1), (5, 3), (33, 8), (100, 10)):
        pts = m.random_points(g2, n, rng, distinct)
        sc = m.random_scalars(g2, n, rng)
        cases.append(case(f"random_n{n}", g2, pts, sc, "G2: uniform scalars < r; replicated bases", check_ref=False))
    pts = m.random_points(g2, 24, rng, 4)
    sc = m.random_scalars(g2, 24, rng)
    sc[0], sc[1], sc[2] = 0, 1, g2.r - 1
    pts[5] = None
    cases.append(case("special_scalars_and_infinity", g2, pts, sc, "G2: 0, 1, r-1 scalars and an infinity base", check_ref=False))
    gg = g2.generator()
    cases.append(case("alternating_pm_generator", g2, [gg if i % 2 == 0 else g2.neg(gg) for i in range(16)], [sc[6]] * 16,
                      "G2: +G, -G with one scalar -> infinity", check_ref=False))
    # BLS12-381 G2 (Fq2 = Fq[u]/(u^2 + 1), b' = 4(1 + u): ARKC bls12_381/src/curves/g2.rs:47-48, fields/fq2.rs:13)
    h2 = m.BLS12_381_G2
    rng = random.Random(0xC0FFEE + 3)
    for n, distinct in ((1, 1), (5, 3), (33, 8), (100, 10)):
        pts = m.random_points(h2, n, rng, distinct)
        sc = m.random_scalars(h2, n, rng)
        cases.append(case(f"random_n{n}", h2, pts, sc, "381 G2: uniform scalars < r; replicated bases", check_ref=False))
    pts = m.random_points(h2, 24, rng, 4)
    sc = m.random_scalars(h2, 24, rng)
    sc[0], sc[1], sc[2] = 0, 1, h2.r - 1
    pts[5] = None
    cases.append(case("special_scalars_and_infinity", h2, pts, sc, "381 G2: 0, 1, r-1 scalars and an infinity base", check_ref=False))
    gg = h2.generator()
    cases.append(case("alternating_pm_generator", h2, [gg if i % 2 == 0 else h2.neg(gg) for i in range(16)], [sc[6]] * 16,
                      "381 G2: +G, -G with one scalar -> infinity", check_ref=False))
    with open(os.path.join(OUT, "msm_vectors.json"), "w") as f:
        json.dump({"generator": "tools/gen_golden.py", "cases": cases}, f, indent=0)
    print("wrote", len(cases), "cases")

    # ---- sizes 2^10 and 2^12: every curve, compact images, cross-checked against everything that can run them ---------------
    large = []
    for curve in (m.BLS12_377_G1, m.BLS12_381_G1, m.BLS12_377_G2, m.BLS12_381_G2):
        for npow, distinct in ((10, 64), (12, 128)):
            n = 1 << npow
            rng = random.Random(0xB16 + 16 * curve.curve_id + npow)
            base = m.random_points(curve, distinct, rng, distinct)
            base[5] = None                                  # an infinity base (replicated n / distinct times)
            base[9] = curve.neg(base[8])                    # a base and its negation meet in buckets
            pts = list(base)
            while len(pts) < n:
                pts.extend(pts[: n - len(pts)])
            sc = m.random_scalars(curve, n, rng)
            sc[0], sc[1], sc[2], sc[3] = 0, 1, curve.r - 1, 2
            sc[8 + distinct] = sc[8]                         # equal scalars on equal bases: the doubling branch of a bucket add
            exp = curve.encode_projective_normalized(curve.msm_pippenger(pts, sc))
            bases_img, scal_img = curve.encode_affine_array(pts), m.encode_scalars(sc)
            assert oracle_c(curve, bases_img, scal_img, n) == exp, (curve.name, n)
            checked = ["pymodel.msm_pippenger", "oracle/msm_oracle.c"]
            if curve.curve_id == 0:
                assert blst377(curve, pts, sc) == exp
                checked.append("oracle/_ref/libblst377.so (reference blst Pippenger)")
                if npow == 10:
                    assert ref377(curve, pts, sc) == exp
                    checked.append("oracle/_ref/libref377.so (reference HostCurve, naive)")
            if curve.curve_id == 1:
                # the reference binary has no infinity input form: run it on the finite pairs only (the dropped ones add nothing)
                keep = [i for i in range(n) if pts[i] is not None]
                r = ref381(curve, [pts[i] for i in keep], [sc[i] for i in keep])
                assert r == exp
                checked.append("oracle/_ref/yrrid381_msm (reference C MSM)")
            large.append({"name": f"large_2^{npow}", "curve": curve.name, "n": n, "distinct": distinct,
                          "note": "bases = the `distinct` records replicated by doubling the vector up to n (P1A yrrid/src/util.rs:15-28); "
                                  "record 5 is infinity, record 9 = -record 8; scalars 0, 1, r-1, 2 lead",
                          "distinct_bases": curve.encode_affine_array(base).hex(), "scalars": scal_img.hex(), "expected": exp.hex(),
                          "checked_against": checked})
            print("large", curve.name, n, "ok:", ", ".join(checked), flush=True)
    with open(os.path.join(OUT, "msm_vectors_large.json"), "w") as f:
        json.dump({"generator": "tools/gen_golden.py", "cases": large}, f, indent=0)

    # literal constants the reference holds (data, not code)
    consts = {
        "source": "SPK ff/bls12-377.hpp:10-25, ff/bls12-381.hpp:10-25 (64-bit limbs, little-endian); "
                  "generators ARKC bls12_377/src/curves/g1.rs:153-159, bls12_381/src/curves/g1.rs:69-73",
        "bls12_377_g1": {
            "P": ["0x8508c00000000001", "0x170b5d4430000000", "0x1ef3622fba094800", "0x1a22d9f300f5138f", "0xc63b05c06ca1493b", "0x01ae3a4617c510ea"],
            "RR": ["0xb786686c9400cd22", "0x0329fcaab00431b1", "0x22a5f11162d6b46d", "0xbfdf7d03827dc3ac", "0x837e92f041790bf9", "0x006dfccb1e914b88"],
            "ONE": ["0x02cdffffffffff68", "0x51409f837fffffb1", "0x9f7db3a98a7d3ff2", "0x7b4e97b76e7c6305", "0x4cf495bf803c84e8", "0x008d6661e2fdf49a"],
            "M0": "0xffffffff",
            "r": ["0x0a11800000000001", "0x59aa76fed0000001", "0x60b44d1e5c37b001", "0x12ab655e9a2ca556"],
            "GX": str(c.gx), "GY": str(c.gy), "B": 1,
        },
        "bls12_381_g1": {
            "P": ["0xb9feffffffffaaab", "0x1eabfffeb153ffff", "0x6730d2a0f6b0f624", "0x64774b84f38512bf", "0x4b1ba7b6434bacd7", "0x1a0111ea397fe69a"],
            "RR": ["0xf4df1f341c341746", "0x0a76e6a609d104f1", "0x8de5476c4c95b6d5", "0x67eb88a9939d83c0", "0x9a793e85b519952d", "0x11988fe592cae3aa"],
            "ONE": ["0x760900000002fffd", "0xebf4000bc40c0002", "0x5f48985753c758ba", "0x77ce585370525745", "0x5c071a97a256ec6d", "0x15f65ec3fa80e493"],
            "M0": "0xfffcfffd",
            "r": ["0xffffffff00000001", "0x53bda402fffe5bfe", "0x3339d80809a1d805", "0x73eda753299d7d48"],
            "GX": str(m.BLS12_381_G1.gx), "GY": str(m.BLS12_381_G1.gy), "B": 4,
        },
    }
    cpath = os.path.join(OUT, "constants.json")
    if os.path.exists(cpath):   # keep what tools/extract_g2_consts.py added (G2 literals)
        old = json.load(open(cpath))
        for k, v in old.items():
            consts.setdefault(k, v)
    with open(cpath, "w") as f:
        json.dump(consts, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
