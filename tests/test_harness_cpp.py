"""The compiled-host side of the boundary: include/mi355_msm.hpp (C++ mirror of the Rust operator API) and the
harness-named shims, exercised by a C++ restatement of the reference's msm_correctness test."""
import ctypes
import os
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "2022-entries_amd")
EXE = os.path.join(ROOT, "tests", "harness_msm_correctness.bin")
EXES = {"377": EXE, "381": os.path.join(ROOT, "tests", "harness_msm_correctness_381.bin")}


def _build():
    """One binary per curve, like the reference's cargo feature (-DFEATURE_BLS12_377 / _381), each against its own shim object."""
    src = os.path.join(ROOT, "tests", "harness_msm_correctness.cpp")
    for cv, exe in EXES.items():
        if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
            subprocess.run(["g++", "-O2", "-std=c++17", f"-DFEATURE_BLS12_{cv}", "-I" + os.path.join(ROOT, "include"), src, "-o", exe,
                            "-L" + PKG, f"-lmi355msm_zprize_{cv}", "-lmi355msm", "-ldl", "-Wl,-rpath," + PKG], check=True)


def test_harness_builds_and_shims_export_reference_names(built):
    _build()
    expect = {
        "libmi355msm_sppark_377.so": ["mult_pippenger_inf"],
        "libmi355msm_sppark_381.so": ["mult_pippenger_inf"],
        "libmi355msm_zprize_377.so": ["mult_pippenger_init", "mult_pippenger_inf"],
        "libmi355msm_zprize_381.so": ["mult_pippenger_init", "mult_pippenger_inf"],
        "libmi355msm_yrrid_377.so": ["MSMAllocContext", "MSMFreeContext", "MSMPreprocessPoints", "MSMRun"],
        "libmi355msm_msm_377.so": ["msm"],
        "libmi355msm_msm_381.so": ["msm"],
    }
    for lib, names in expect.items():
        out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(PKG, lib)], capture_output=True, text=True, check=True).stdout
        for n in names:
            assert f" T {n}" in out, (lib, n)


@pytest.mark.gpu
@pytest.mark.parametrize("cv", ["377", "381"])
@pytest.mark.parametrize("npow,batches", [(10, 4), (15, 2)])
def test_msm_correctness_cpp(built, npow, batches, cv):
    _build()
    r = subprocess.run([EXES[cv], os.path.join(ROOT, "oracle", "liboracle.so"), str(npow), str(batches)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"curve={0 if cv == '377' else 1} " in r.stdout and ": ok" in r.stdout


class _RustError(ctypes.Structure):
    _fields_ = [("code", ctypes.c_int), ("message", ctypes.c_void_p)]


class _RustContext(ctypes.Structure):
    _fields_ = [("context", ctypes.c_void_p)]


def _inputs(ea, cid, n, seed, batches=1):
    import numpy as np

    curve = {0: "bls12_377_g1", 1: "bls12_381_g1"}[cid]
    bases = ea.generate_points(n, distinct=min(n, 1 << 11), seed=seed, curve=curve)
    bases[3, 96] = 1                                   # the baseline harness plants a point at infinity (P1A 6block/src/util.rs:26)
    rng = np.random.default_rng(seed)
    limbs = rng.integers(0, 1 << 64, size=(batches * n, 4), dtype=np.uint64)
    limbs[:, 3] %= np.uint64(0x12ab655e9a2ca556)       # below both group orders
    return bases, limbs.view(np.uint8).reshape(batches * n, 32)


@pytest.mark.gpu
@pytest.mark.parametrize("cv,cid", [("377", 0), ("381", 1)])
@pytest.mark.parametrize("npow", [12, 20])
def test_north_star_msm_symbol_on_the_gpu(built, oracle, ea, cv, cid, npow):
    """The literal entry point BASELINE.json names -- `msm(out, bases, scalars, n)` of libmi355msm_msm_{377,381}.so
    (csrc/shims/north_star_msm.c; stands for SPK poc/blst-cuda/cuda/pippenger_inf.cu:28-35) -- EXECUTED, against the oracle."""
    from conftest import oracle_msm_np

    lib = ctypes.CDLL(os.path.join(PKG, f"libmi355msm_msm_{cv}.so"))
    lib.msm.restype = _RustError
    lib.msm.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    n = 1 << npow
    bases, sc = _inputs(ea, cid, n, seed=100 + npow + cid)
    out = ctypes.create_string_buffer(144)
    err = lib.msm(out, bases.ctypes.data, sc.ctypes.data, n)
    assert err.code == 0 and not err.message
    assert out.raw == oracle_msm_np(oracle, cid, bases, sc, n)
    # n = 0: the point at infinity (1, 1, 0), no error; a null operand with n > 0: an error WITH a message (SPK util/rusterror.h:15-27)
    err = lib.msm(out, None, None, 0)
    assert err.code == 0 and out.raw[96:] == bytes(48)
    err = lib.msm(out, None, sc.ctypes.data, 5)
    assert err.code != 0 and err.message
    ctypes.CDLL(None).free(ctypes.c_void_p(err.message))


@pytest.mark.gpu
def test_sppark_and_zprize_names_on_bls12_381(built, oracle, ea):
    """The 381 builds of the harness-named shims, executed: the 5-argument stateless mult_pippenger_inf
    (SPK poc/blst-cuda/cuda/pippenger_inf.cu:28-35) and the context pair mult_pippenger_init / 7-argument mult_pippenger_inf
    (P1A 6block/cuda/pippenger_inf.cu:50-53, 87-92) with several batches."""
    import numpy as np

    from conftest import oracle_msm_np

    n, batches = 1 << 13, 3
    bases, sc = _inputs(ea, 1, n, seed=381, batches=batches)
    s = ctypes.CDLL(os.path.join(PKG, "libmi355msm_sppark_381.so"))
    s.mult_pippenger_inf.restype = _RustError
    s.mult_pippenger_inf.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    out = ctypes.create_string_buffer(144)
    assert s.mult_pippenger_inf(out, bases.ctypes.data, n, sc.ctypes.data, 104).code == 0
    exp0 = oracle_msm_np(oracle, 1, bases, np.ascontiguousarray(sc[:n]), n)
    assert out.raw == exp0
    z = ctypes.CDLL(os.path.join(PKG, "libmi355msm_zprize_381.so"))
    z.mult_pippenger_init.restype = _RustError
    z.mult_pippenger_init.argtypes = [ctypes.POINTER(_RustContext), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]
    z.mult_pippenger_inf.restype = _RustError
    z.mult_pippenger_inf.argtypes = [ctypes.POINTER(_RustContext), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                     ctypes.c_void_p, ctypes.c_size_t]
    rc = _RustContext(None)
    assert z.mult_pippenger_init(ctypes.byref(rc), bases.ctypes.data, n, 104).code == 0 and rc.context
    outs = ctypes.create_string_buffer(144 * batches)
    assert z.mult_pippenger_inf(ctypes.byref(rc), outs, bases.ctypes.data, n, batches, sc.ctypes.data, 104).code == 0
    assert outs.raw[:144] == exp0
    for b in range(1, batches):
        assert outs.raw[144 * b:144 * (b + 1)] == oracle_msm_np(oracle, 1, bases, np.ascontiguousarray(sc[b * n:(b + 1) * n]), n)
    err = z.mult_pippenger_inf(ctypes.byref(rc), outs, bases.ctypes.data, n, 0, sc.ctypes.data, 104)     # batches = 0: error with a message
    assert err.code != 0 and err.message
    ctypes.CDLL(None).free(ctypes.c_void_p(err.message))
    # like the reference, the harness context is never freed by the harness (P1A 6block/cuda/pippenger_inf.cu:55); free it here
    ea.load_library().mi355_msm_destroy(rc.context)


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
