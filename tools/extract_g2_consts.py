#!/usr/bin/env python3
"""Pull the BLS12-377 and BLS12-381 G2 literals the reference holds -- generator coordinates, the twist coefficient b' and the Fq2
non-residue -- out of its sources into tests/golden/constants.json (data, not code; run in the build container only:
/root/reference does not exist on the GPU box).

  ARKC bls12_377/src/curves/g2.rs:47-50   COEFF_B = (0, 1551...906)
  ARKC bls12_377/src/curves/g2.rs:61-78   G2_GENERATOR_{X,Y}_{C0,C1}
  ARKC bls12_377/src/fields/fq2.rs:13     NONRESIDUE = -5
  ARKC bls12_381/src/curves/g2.rs:47-48   COEFF_B = (g1 COEFF_B, g1 COEFF_B) = (4, 4)   (g1.rs:36-37)
  ARKC bls12_381/src/curves/g2.rs:74-91   G2_GENERATOR_{X,Y}_{C0,C1}
  ARKC bls12_381/src/fields/fq2.rs:13     NONRESIDUE = -1
  ARKC bls12_377/src/curves/g2.rs:17-34, bls12_381/src/curves/g2.rs:22-40   COFACTOR (u64 limbs) and COFACTOR_INV = COFACTOR^-1 mod r
  ARKC bls12_377/src/fields/fq6.rs:15-22, fq12.rs:15-22                      Fq2::NONRESIDUE-of-the-tower powers u^((q-1)/3), u^((q-1)/6)
                                                                             (Frobenius coefficients): known answers for Fq2 powers with beta = -5
"""
import json
import os
import re

ARKC = "/root/reference/open-division/prize4-msm-wasm/snarkify/zprize-prize4-15ac8c55-arkworks-curves/bls12_377/src"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    g2 = open(os.path.join(ARKC, "curves", "g2.rs")).read()
    fq2 = open(os.path.join(ARKC, "fields", "fq2.rs")).read()
    out = {}
    for name in ("X_C0", "X_C1", "Y_C0", "Y_C1"):
        mm = re.search(r"pub const G2_GENERATOR_%s: Fq = MontFp!\(\"(\d+)\"\);" % name, g2)
        out["G" + name.replace("_C", "")] = mm.group(1)
    mm = re.search(r"const COEFF_B: Fq2 = Fq2::new\(\s*Fq::ZERO,\s*MontFp!\(\"(\d+)\"\),\s*\);", g2)
    out["B0"], out["B1"] = "0", mm.group(1)
    mm = re.search(r"const NONRESIDUE: Fq = MontFp!\(\"(-?\d+)\"\);", fq2)
    out["NONRESIDUE"] = mm.group(1)
    # BLS12-381 G2
    src381 = ARKC.replace("bls12_377", "bls12_381")
    g2b = open(os.path.join(src381, "curves", "g2.rs")).read()
    g1b = open(os.path.join(src381, "curves", "g1.rs")).read()
    fq2b = open(os.path.join(src381, "fields", "fq2.rs")).read()
    out381 = {}
    for name in ("X_C0", "X_C1", "Y_C0", "Y_C1"):
        mm = re.search(r"pub const G2_GENERATOR_%s: Fq = MontFp!\(\"(\d+)\"\);" % name, g2b)
        out381["G" + name.replace("_C", "")] = mm.group(1)
    assert re.search(r"const COEFF_B: Fq2 = Fq2::new\(g1::Parameters::COEFF_B, g1::Parameters::COEFF_B\);", g2b)
    mm = re.search(r"const COEFF_B: Fq = MontFp!\(\"(\d+)\"\);", g1b)
    out381["B0"] = out381["B1"] = mm.group(1)
    mm = re.search(r"const NONRESIDUE: Fq = MontFp!\(\"(-?\d+)\"\);", fq2b)
    out381["NONRESIDUE"] = mm.group(1)
    # cofactors (both curves) and two Frobenius coefficients (BLS12-377): round 6, the known answers of the G2 cofactor test
    def cof(src, dst):
        mm = re.search(r"const COFACTOR: &'static \[u64\] = &\[(.*?)\];", src, flags=re.S)
        limbs = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]+)", mm.group(1))]
        dst["COFACTOR"] = str(sum(v << (64 * i) for i, v in enumerate(limbs)))
        head = src[:mm.start()]
        docs = re.findall(r"\d{100,}", head[head.rindex("/// COFACTOR ="):])
        assert dst["COFACTOR"] in docs, "the limbs and the doc comment of COFACTOR disagree"
        dst["COFACTOR_INV"] = re.search(r"const COFACTOR_INV: Fr =\s*MontFp!\(\"(\d+)\"\);", src).group(1)
    cof(g2, out)
    cof(g2b, out381)
    fq6 = open(os.path.join(ARKC, "fields", "fq6.rs")).read()
    fq12 = open(os.path.join(ARKC, "fields", "fq12.rs")).read()
    out["FROB6_C1_1"] = re.search(r"NONRESIDUE\^\(\(\(q\^1\) - 1\) / 3\)\s*Fq2::new\(\s*MontFp!\(\"(\d+)\"\),\s*Fq::ZERO,", fq6).group(1)
    out["FROB12_C1_1"] = re.search(r"NONRESIDUE\^\(\(\(q\^1\) - 1\) / 6\)\s*Fq2::new\(\s*MontFp!\(\"(\d+)\"\),\s*Fq::ZERO,", fq12).group(1)
    path = os.path.join(ROOT, "tests", "golden", "constants.json")
    data = json.load(open(path))
    data["bls12_377_g2"] = out
    data["bls12_381_g2"] = out381
    data["source_g2"] = ("ARKC bls12_377/src/curves/g2.rs:47-50 (COEFF_B), :61-78 (generator), fields/fq2.rs:13 (NONRESIDUE); "
                         "bls12_381/src/curves/g2.rs:47-48 + g1.rs:36-37 (COEFF_B), g2.rs:74-91 (generator), fields/fq2.rs:13 (NONRESIDUE); "
                         "COFACTOR / COFACTOR_INV: bls12_377/src/curves/g2.rs:17-34, bls12_381/src/curves/g2.rs:22-40; FROB6_C1_1 / FROB12_C1_1 = u^((q-1)/3), "
                         "u^((q-1)/6) (c1 = 0): bls12_377/src/fields/fq6.rs:18-22, fq12.rs:18-22; "
                         "decimal literals, normal form; extracted by tools/extract_g2_consts.py")
    with open(path, "w") as f:
        json.dump(data, f, indent=1)
        f.write("\n")
    print(json.dumps({"bls12_377_g2": out, "bls12_381_g2": out381}, indent=1))


if __name__ == "__main__":
    main()
