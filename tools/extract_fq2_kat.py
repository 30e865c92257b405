#!/usr/bin/env python3
"""Extract the BLS12-381 Fq2 known-answer vectors the reference's tests hold (data only: 6-limb integers) into
tests/golden/fq2_kat_bls12_381.json.  Source: ARKC bls12_381/src/fields/tests.rs:1232-1392
(test_fq2_squaring, test_fq2_mul, test_fq2_inverse).  Run in this container (needs /root/reference)."""
import json
import os
import re

REF = "/root/reference/open-division/prize4-msm-wasm/snarkify/zprize-prize4-15ac8c55-arkworks-curves/bls12_381/src/fields/tests.rs"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(REF).read()


def section(name, nxt):
    a = src.index("fn %s()" % name)
    return src[a:src.index("fn %s()" % nxt, a)]


def elems(text):
    """all BigInt::new([l0..l5]) literals, in order, as Python ints"""
    out = []
    for m in re.finditer(r"BigInt::new\(\[([^\]]*)\]\)", text):
        limbs = [int(x.strip(), 16) for x in m.group(1).split(",") if x.strip()]
        assert len(limbs) == 6
        out.append(sum(l << (64 * i) for i, l in enumerate(limbs)))
    return out


sq = elems(section("test_fq2_squaring", "test_fq2_mul"))
mu = elems(section("test_fq2_mul", "test_fq2_inverse"))
iv = elems(section("test_fq2_inverse", "test_fq2_addition"))
assert len(sq) == 4 and len(mu) == 6 and len(iv) == 4
kat = {
    "source": "ARKC bls12_381/src/fields/tests.rs:1232-1392 (Fq2 = Fq[u]/(u^2+1); values in normal form)",
    "square": [{"a": [str(sq[0]), str(sq[1])], "out": [str(sq[2]), str(sq[3])]}],
    "mul": [{"a": [str(mu[0]), str(mu[1])], "b": [str(mu[2]), str(mu[3])], "out": [str(mu[4]), str(mu[5])]}],
    "inverse": [{"a": [str(iv[0]), str(iv[1])], "out": [str(iv[2]), str(iv[3])]}],
}
with open(os.path.join(ROOT, "tests", "golden", "fq2_kat_bls12_381.json"), "w") as f:
    json.dump(kat, f, indent=1)
print("ok")
