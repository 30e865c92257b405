"""GPU: the HIP path against the reference's OWN code, built here from its sources by oracle/Makefile (oracle/_ref/, binaries
that travel to the GPU box; /root/reference itself is not read at run time): yrrid's host BLS12-377 XYZZ code driving a naive
MSM (CMB yrrid-ff-ec/HostCurve.cpp) and yrrid's C BLS12-381 G1 MSM (open-division/prize4-msm-wasm/yrrid/C/MSM.c).  This is
"outputs of the reference itself run here" compared DIRECTLY with the kernels -- not through the restatement."""
import ctypes
import os
import random

import pytest

import pymodel as m
from conftest import ROOT
from test_oracle import REF377, REF381, _run_ref381

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not os.path.exists(REF377), reason="oracle/_ref not built")
@pytest.mark.parametrize("twisted_edwards", [1, 0])
def test_hip_377_equals_reference_hostcurve_msm(ea, twisted_edwards):
    ref = ctypes.CDLL(REF377)
    ref.ref377_msm_naive.restype = ctypes.c_int
    c = m.BLS12_377_G1
    rng = random.Random(4077)
    n = 1500
    pts = m.random_points(c, n, rng, 97)
    pts[11] = None
    sc = m.random_scalars(c, n, rng)
    sc[5], sc[6] = 0, 1
    bases, scalars = c.encode_affine_array(pts), m.encode_scalars(sc)
    out = ctypes.create_string_buffer(144)
    inf = ref.ref377_msm_naive(bases, ctypes.c_size_t(104), scalars, ctypes.c_size_t(n), out)
    expect = c.encode_projective_normalized(None) if inf else out.raw
    ctx = ea.MultiScalarMultContext("bls12_377_g1")
    ctx.set_option("twisted_edwards", twisted_edwards)
    ctx.set_bases(bases)
    assert ctx.query("twisted_edwards") == twisted_edwards
    assert ctx.run(scalars)[0] == expect
    ctx.close()


@pytest.mark.skipif(not os.path.exists(REF381), reason="oracle/_ref not built")
def test_hip_381_equals_reference_c_msm(ea):
    c = m.BLS12_381_G1
    rng = random.Random(4381)
    n = 4096
    pts = m.random_points(c, n, rng, 256)
    sc = m.random_scalars(c, n, rng)
    ref_pt = _run_ref381(c, pts, sc)
    got = ea.msm(c.encode_affine_array(pts), m.encode_scalars(sc), "bls12_381_g1")
    assert got == c.encode_projective_normalized(ref_pt)
