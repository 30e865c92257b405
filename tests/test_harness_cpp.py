"""The compiled-host side of the boundary: include/mi355_msm.hpp (C++ mirror of the Rust operator API) and the
harness-named shims, exercised by a C++ restatement of the reference's msm_correctness test."""
import os
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "2022-entries_amd")
EXE = os.path.join(ROOT, "tests", "harness_msm_correctness.bin")


def _build():
    src = os.path.join(ROOT, "tests", "harness_msm_correctness.cpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", EXE,
                        "-L" + PKG, "-lmi355msm_zprize_377", "-lmi355msm", "-ldl", "-Wl,-rpath," + PKG], check=True)


def test_harness_builds_and_shims_export_reference_names(built):
    _build()
    expect = {
        "libmi355msm_sppark_377.so": ["mult_pippenger_inf"],
        "libmi355msm_sppark_381.so": ["mult_pippenger_inf"],
        "libmi355msm_zprize_377.so": ["mult_pippenger_init", "mult_pippenger_inf"],
        "libmi355msm_zprize_381.so": ["mult_pippenger_init", "mult_pippenger_inf"],
        "libmi355msm_yrrid_377.so": ["MSMAllocContext", "MSMFreeContext", "MSMPreprocessPoints", "MSMRun"],
    }
    for lib, names in expect.items():
        out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(PKG, lib)], capture_output=True, text=True, check=True).stdout
        for n in names:
            assert f" T {n}" in out, (lib, n)


@pytest.mark.gpu
@pytest.mark.parametrize("npow,batches", [(10, 4), (15, 2)])
def test_msm_correctness_cpp(built, npow, batches):
    _build()
    r = subprocess.run([EXE, os.path.join(ROOT, "oracle", "liboracle.so"), str(npow), str(batches)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_yrrid_and_sppark_names(built, oracle):
    """MSMAllocContext/.../MSMRun and the 5-argument mult_pippenger_inf, through ctypes."""
    import ctypes

    import numpy as np

    import entries_amd as ea
    from conftest import oracle_msm_np

    n = 2048
    bases = ea.generate_points(n, distinct=128, seed=9)
    rng = np.random.default_rng(1)
    sc = rng.integers(0, 256, size=(2 * n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x0F
    y = ctypes.CDLL(os.path.join(PKG, "libmi355msm_yrrid_377.so"))
    y.MSMAllocContext.restype = ctypes.c_void_p
    y.MSMFreeContext.argtypes = [ctypes.c_void_p]
    y.MSMPreprocessPoints.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    y.MSMRun.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    ctx = y.MSMAllocContext(1 << 26, 16)
    assert ctx and y.MSMPreprocessPoints(ctx, bases.ctypes.data, n) == 0
    out = ctypes.create_string_buffer(288)
    assert y.MSMRun(ctx, out, sc.ctypes.data, 2 * n) == 0
    for b in range(2):
        assert out.raw[144 * b:144 * (b + 1)] == oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(sc[b * n:(b + 1) * n]), n)
    assert y.MSMRun(ctx, out, sc.ctypes.data, n + 1) != 0      # not a whole number of batches: sticky error
    assert y.MSMRun(ctx, out, sc.ctypes.data, n) != 0
    y.MSMFreeContext(ctx)

    class RustError(ctypes.Structure):
        _fields_ = [("code", ctypes.c_int), ("message", ctypes.c_void_p)]

    s = ctypes.CDLL(os.path.join(PKG, "libmi355msm_sppark_377.so"))
    s.mult_pippenger_inf.restype = RustError
    s.mult_pippenger_inf.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    out1 = ctypes.create_string_buffer(144)
    err = s.mult_pippenger_inf(out1, bases.ctypes.data, n, sc.ctypes.data, 104)
    assert err.code == 0
    assert out1.raw == oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(sc[:n]), n)
