"""CPU: the plumbing of the C-ABI sharded context (include/mi355_msm.h: mi355_msm_create_sharded / _create_env) that needs no
GPU -- slice bounds (C twin of dist.shard_bounds), slice-and-fold against the oracle, MI355_MSM_DEVICES parsing, and that a
sharded context fails loudly without a device (no CPU fallback)."""
import ctypes
import os
import random
import subprocess
import sys

import pytest

import pymodel as m
from conftest import ROOT, oracle_msm


def _c_bounds(lib, n, G, g):
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    err = lib.mi355_msm_shard_bounds(n, G, g, ctypes.byref(lo), ctypes.byref(hi))
    assert err.code == 0
    return lo.value, hi.value


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 1000, (1 << 28), (1 << 28) + 5])
@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_shard_bounds_c_matches_python_and_partitions(ea, n, G):
    lib = ea.load_library()
    prev = 0
    for g in range(G):
        lo, hi = _c_bounds(lib, n, G, g)
        assert (lo, hi) == ea.shard_bounds(n, G, g)
        assert lo == prev and lo <= hi <= n      # contiguous, disjoint, in order
        prev = hi
    assert prev == n
    if n == 1 << 28 and G == 8:
        assert _c_bounds(lib, n, G, 3) == (3 << 25, 4 << 25)   # BASELINE config 4: 8 shards of 2^25
    err = lib.mi355_msm_shard_bounds(10, 2, 2, ctypes.byref(ctypes.c_size_t()), ctypes.byref(ctypes.c_size_t()))
    assert err.code != 0 and err.message
    ctypes.CDLL(None).free(ctypes.c_void_p(err.message))


@pytest.mark.parametrize("curve,n,G", [(m.BLS12_377_G1, 203, 8), (m.BLS12_381_G1, 50, 3), (m.BLS12_377_G1, 5, 8)])
def test_slice_and_fold_equals_whole(ea, oracle, curve, n, G):
    """What sharded_run does after the shards return: partial per slice (here: the oracle), then mi355_msm_fold."""
    rng = random.Random(n * G)
    pts = m.random_points(curve, n, rng, max(1, n // 3))
    sc = m.random_scalars(curve, n, rng)
    bases, scalars = curve.encode_affine_array(pts), m.encode_scalars(sc)
    lib = ea.load_library()
    partials = []
    for g in range(G):
        lo, hi = _c_bounds(lib, n, G, g)
        partials.append(oracle_msm(oracle, curve.curve_id, bases[lo * 104:hi * 104], scalars[lo * 32:hi * 32], hi - lo))
    assert ea.fold_partials(partials, curve.name) == oracle_msm(oracle, curve.curve_id, bases, scalars, n)


def _create_env(ea, value):
    env = dict(os.environ)
    if value is None:
        env.pop("MI355_MSM_DEVICES", None)
    else:
        env["MI355_MSM_DEVICES"] = value
    code = ("import sys; sys.path.insert(0, %r); import entries_amd as ea\n"
            "try:\n    ea.MultiScalarMultContext.from_env('bls12_377_g1'); print('OK')\n"
            "except ea.MsmError as e:\n    print('ERR', e.code, e.message)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1]


def test_devices_env_is_parsed_before_anything_touches_a_device(ea):
    for bad in ("0,,1", "a", "3-1", "0-", "0,1x"):
        line = _create_env(ea, bad)
        assert line.startswith("ERR") and "MI355_MSM_DEVICES" in line, (bad, line)


def test_sharded_context_fails_loudly_without_gpu(ea):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ea.MsmError) as ei:
        ea.MultiScalarMultContext("bls12_377_g1", devices=[0, 0])
    assert "no HIP device" in ei.value.message
    for value in ("0,1", "0-7", None):
        line = _create_env(ea, value)
        assert line.startswith("ERR") and "no HIP device" in line, (value, line)
    with pytest.raises(ea.MsmError):
        ea.MultiScalarMultContext("bls12_377_g1", devices=[])
