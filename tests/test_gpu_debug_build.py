"""GPU: the -DMSM_DEBUG build of the engine (2022-entries_amd/libmi355msm_debug.so, built by build.py beside the product from the
same kernel objects) -- device-side invariant checks after every grouping level and after the accumulation (csrc/partition.hpp:
segment tables contiguous and ending at the entry total, only unresolved key bits left in a level's entries, sorted keys
non-decreasing and in range, values naming bases of the chunk, entry count == an independent count of the non-zero digits, slot
keys KEY_NONE or valid).  The reference keeps such a self-check, disabled, in its partition (CMB Partition4096.cu:419-432).

Run in a subprocess so that the product library of this test session and the debug build never share a process."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.environ["REPO"])
import entries_amd as ea
lib = ea.load_library()
assert b"+debug-invariants" in lib.mi355_msm_version(), lib.mi355_msm_version()
oracle = ctypes.CDLL(os.path.join(os.environ["REPO"], "oracle", "liboracle.so"))
oracle.oracle_msm.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
out = {}
for curve, cid, npow in (("bls12_377_g1", 0, 20), ("bls12_381_g1", 1, 20), ("bls12_377_g2", 2, 18), ("bls12_381_g2", 3, 16)):
    n = 1 << npow
    bases = ea.generate_points(n, distinct=1 << 11, seed=cid + 1, curve=curve)
    rng = np.random.default_rng(cid)
    limbs = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] %= np.uint64(0x12ab655e9a2ca556)
    sc = limbs.view(np.uint8).reshape(n, 32)
    sc[5] = 0
    bases[9, ea.affine_stride(curve) - 8] = 1
    results = {}
    for opts in ({}, {"max_chunk": n // 3 + 1}, {"precompute": 1}, {"assume_subgroup": 1}):
        if opts.get("precompute") and cid >= 2:
            continue
        ctx = ea.MultiScalarMultContext(curve)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_bases(bases)
        got = ctx.run(sc)[0]
        checks = ctx.query("debug_checks")
        ctx.close()
        assert checks >= 3, (curve, opts, checks)     # level 1, the sorted output + digit count, the slot keys: per chunk
        results[json.dumps(opts)] = [got.hex(), checks]
    small = 1 << 13     # a prefix the oracle finishes quickly: the debug build's results are the product's results
    exp = ctypes.create_string_buffer(ea.projective_bytes(curve))
    assert oracle.oracle_msm(cid, bases.ctypes.data, ea.affine_stride(curve), sc.ctypes.data, small, exp, 0) == 0
    assert ea.msm(bases[:small], sc[:small], curve) == exp.raw, curve
    assert len({v[0] for v in results.values()}) == 1, (curve, "options changed the result")
    out[curve] = {k: v[1] for k, v in results.items()}
print("DEBUG_BUILD_OK " + json.dumps(out))
"""


def _run(extra_env):
    env = dict(os.environ, REPO=ROOT, MI355_MSM_LIBRARY="libmi355msm_debug.so", **extra_env)
    return subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)


def test_debug_build_holds_its_invariants_on_all_curves(built):
    assert os.path.exists(os.path.join(ROOT, "2022-entries_amd", "libmi355msm_debug.so")), "build.py builds it beside the product"
    r = _run({})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("DEBUG_BUILD_OK ")]
    assert line, r.stdout[-2000:]
    checks = json.loads(line[0][len("DEBUG_BUILD_OK "):])
    assert set(checks) == {"bls12_377_g1", "bls12_381_g1", "bls12_377_g2", "bls12_381_g2"}
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "r05_debug_build_checks.json"), "w") as f:
        json.dump(checks, f, indent=1)


def test_debug_build_catches_a_planted_violation(built):
    """MI355_MSM_DEBUG_CORRUPT=1 (debug build only) swaps the keys of two neighbouring sorted entries before the checks run: the
    run must FAIL with the invariant named, not return a point."""
    r = _run({"MI355_MSM_DEBUG_CORRUPT": "1"})
    assert r.returncode != 0
    assert "MSM_DEBUG: sorted keys decrease" in (r.stdout + r.stderr), (r.stdout + r.stderr)[-3000:]


def test_product_build_has_no_debug_checks(ea):
    """The shipped library is not the debug build: no check ever runs in it."""
    assert b"+debug" not in ea.load_library().mi355_msm_version()
    bases = ea.generate_points(4096, distinct=64, seed=3)
    ctx = ea.multi_scalar_mult_init(bases)
    ctx.run(bytes(4096 * 32))
    assert ctx.query("debug_checks") == 0
    ctx.close()
