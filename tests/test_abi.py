"""CPU: the drop-in boundary.  The HIP library loads, exports every symbol include/mi355_msm.h declares, reports errors
the sppark way (RustError by value, message always set), and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import random
import re

import pytest

import pymodel as m
from conftest import ROOT


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "mi355_msm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_msm\w*)\s*\(", src)))


def test_header_symbols_exported(ea):
    lib = ea.load_library()
    names = _declared_functions()
    assert "mi355_msm_run_device" in names and "mi355_msm" in names and len(names) >= 11
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/mi355_msm.h but not exported"
    assert b"gfx950" in lib.mi355_msm_version()


def test_library_embeds_gfx950_code_object(ea):
    blob = open(ea.library_path(), "rb").read()
    assert b"gfx950" in blob and b"k_accumulate" in blob and b"k_bucket_reduce" in blob


def test_no_gpu_fails_loudly(ea):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    c = m.BLS12_377_G1
    pts = m.random_points(c, 4, random.Random(1))
    with pytest.raises(ea.MsmError) as ei:
        ea.msm(c.encode_affine_array(pts), m.encode_scalars([1, 2, 3, 4]))
    assert ei.value.code != 0 and "no HIP device" in ei.value.message
    with pytest.raises(ea.MsmError):
        ea.multi_scalar_mult_init(c.encode_affine_array(pts))


def test_argument_errors_have_messages(ea):
    lib = ea.load_library()
    out = ctypes.create_string_buffer(144)
    err = lib.mi355_msm_fold(7, out, out, 1)
    assert err.code != 0 and err.message
    assert b"unknown curve" in ctypes.string_at(err.message)
    ctypes.CDLL(None).free(ctypes.c_void_p(err.message))
    with pytest.raises(ValueError):
        ea.fold_partials([b"\x00" * 100])


@pytest.mark.parametrize("curve", [m.BLS12_377_G1, m.BLS12_381_G1])
def test_fold_is_host_side_group_addition(ea, curve):
    """mi355_msm_fold: the multi-GPU combine; any-Z Jacobian inputs, doubling and cancellation included."""
    rng = random.Random(3)
    pts = m.random_points(curve, 5, rng)
    enc = curve.encode_projective_normalized
    assert ea.fold_partials([enc(p) for p in pts] + [enc(None)], curve.name) == enc(
        curve.add(curve.add(curve.add(pts[0], pts[1]), curve.add(pts[2], pts[3])), pts[4]))
    assert ea.fold_partials([enc(pts[0]), enc(pts[0])], curve.name) == enc(curve.mul(2, pts[0]))
    assert ea.fold_partials([enc(pts[0]), enc(curve.neg(pts[0]))], curve.name) == enc(None)
    assert ea.fold_partials([], curve.name) == enc(None)
    # a non-normalised Jacobian triple (X z^2, Y z^3, z) is the same point
    z = rng.randrange(2, curve.p)
    x, y = pts[1]
    raw = b"".join(((v * m.R) % curve.p).to_bytes(48, "little") for v in ((x * z * z) % curve.p, (y * z * z * z) % curve.p, z))
    assert ea.fold_partials([raw], curve.name) == enc(pts[1])


@pytest.mark.parametrize("curve", [m.BLS12_377_G1, m.BLS12_381_G1, m.BLS12_377_G2, m.BLS12_381_G2], ids=lambda c: c.name)
def test_generate_points(ea, curve):
    """The synthetic generator mirrors the reference harness: distinct subgroup points, replicated by doubling.  On-curve
    (with the reference's b / b') and order r also pin the generator constants the engine starts from (field_consts.inc)."""
    arr = ea.generate_points(64, distinct=16, seed=5, curve=curve.name)
    stride, cb = curve.affine_stride, curve.coord_bytes
    assert arr.shape == (64, stride)
    pts = [curve.decode_affine(arr[i].tobytes()) for i in range(64)]
    assert all(curve.on_curve(P) and P is not None for P in pts[:16])
    key = (lambda P: P) if curve.ext == 1 else (lambda P: (P[0].c0, P[0].c1, P[1].c0, P[1].c1))
    assert len({key(P) for P in pts[:16]}) == 16
    assert all(curve.mul(curve.r, P) is None for P in pts[:3])
    assert pts[16:32] == pts[:16] and pts[32:] == pts[:32]
    assert (arr[:, 2 * cb:] == 0).all()


def test_python_mirror_argument_checks(ea):
    with pytest.raises(ValueError):
        ea.msm(b"", b"", "no_such_curve")
    assert ea.shard_bounds(10, 4, 0) == (0, 3) and ea.shard_bounds(10, 4, 3) == (9, 10)
    covered = []
    for r in range(8):
        lo, hi = ea.shard_bounds(1 << 20, 8, r)
        covered.append((lo, hi))
    assert covered[0][0] == 0 and covered[-1][1] == 1 << 20 and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))


def test_execution_plan_is_sane_and_terminates(ea):
    """mi355_msm_plan (host arithmetic): for every size class and every tuning knob the fragment merge shrinks to one lane,
    windows cover 256-bit scalars plus the signed-digit carry, and sort entries stay below 2^32."""
    for curve in ("bls12_377_g1", "bls12_381_g1", "bls12_377_g2", "bls12_381_g2"):
        for npow in (0, 1, 5, 10, 16, 20, 24, 26):
            for pre in (False, True):
                p = ea.plan(1 << npow, curve, precompute=pre)
                assert p["window_bits"] * p["windows"] >= 257
                assert p["entries"] == p["windows"] << npow and p["entries"] < 1 << 32
                assert p["bucket_windows"] == (1 if pre else p["windows"])
                assert p["lanes"] * p["lane_entries"] >= p["entries"]
                assert 1 <= p["reduce_launches"] <= 12 and p["merge_launches"] <= 40
                assert (p["windows"] << (p["window_bits"] - 1) if not pre else 1 << (p["window_bits"] - 1)) < 1 << p["key_bits"]
    # BASELINE.json sizes: the canonical 2^26 G1 problem fits comfortably in one MI355X (288 GB)
    p = ea.plan(1 << 26)
    assert p["window_bits"] in (20, 21, 22) and p["work_bytes"] < 40 << 30
    assert ea.plan(1 << 26, precompute=True)["window_bits"] in (22, 23, 24)
    for seg in (4, 5, 7, 64, 4096):
        for k in (1, 3, 4, 1000):
            for n in (1, 2, 3, 1000, 12345, 1 << 20):
                assert ea.plan(n, lane_entries=k, seg_entries=seg)["merge_launches"] <= 64
    with pytest.raises(ea.MsmError):
        ea.plan(100, seg_entries=2)
    with pytest.raises(ea.MsmError):
        ea.plan(100, window_bits=99)


def test_plan_with_table_levels(ea):
    """The plan of a context with "table_levels" = k (the query used to assume a level per window): k levels over
    ceil(windows / k) bucket sets, levels rounded so that none is empty (13 windows in 6 levels are 3 sets x 5 levels), the reference's
    own shape -- 6 levels, 2 bucket sets (CMB PrecomputePoints.cu:10-39, MSM.cu:380-383) -- at 2^26 pairs, and less work memory for
    fewer bucket sets."""
    for curve in ("bls12_377_g1", "bls12_381_g1", "bls12_377_g2"):
        for npow in (14, 20, 26):
            every = ea.plan(1 << npow, curve, precompute=True)
            for k in (2, 3, 6, 40):
                p = ea.plan(1 << npow, curve, precompute=True, table_levels=k)
                levels = -(-p["windows"] // p["bucket_windows"])
                assert p["window_bits"] * p["windows"] >= 257 and levels <= k and p["bucket_windows"] == -(-p["windows"] // min(k, p["windows"]))
                assert (p["bucket_windows"] << (p["window_bits"] - 1)) < 1 << p["key_bits"]
                if k == 40:
                    assert p == every          # more levels than windows: a level per window
    p = ea.plan(1 << 26, precompute=True, table_levels=6)
    assert p["window_bits"] == 23 and p["windows"] == 12 and p["bucket_windows"] == 2
    with pytest.raises(ValueError):
        ea.plan(100, precompute=True, table_levels=1)


def test_bench_telemetry_never_raises_without_a_gpu():
    """bench.py samples the GPU's clock and power in a thread during the timed loop (sysfs hwmon, then amdsmi); on a box without
    either -- this container -- it must say so in `telemetry_source` and report None, never fail the line."""
    import bench

    t = bench.Telemetry(0)
    t.start()
    r = t.stop()
    assert set(r) == {"clock_MHz_mean", "clock_MHz_min", "clock_MHz_max", "power_W_mean", "power_W_max", "samples"}
    if t.source is None:
        assert r["clock_MHz_mean"] is None and r["samples"] == 0 and t.describe().startswith("unavailable")
    probe = bench.ark_ec_probe(10)
    assert probe["available"] in (False, True) and ("probe" in probe or "error" in probe or probe.get("kind") == "ark-ec")
    assert len(bench.kernel_source_sha16()) == 16


def test_window_sizes_the_anchored_window_moves(ea):
    """The window model with the anchored window (csrc/msm_engine.hip anchor_min_c; measured: profiles/r06_ab_anchor.txt, guarded on the GPU by
    tests/test_gpu_window_model.py): BLS12-377 moves only to c = 21 (252 = 12 x 21) -- 2^26 pairs --, never to c = 18 (measured to lose at
    2^23 and 2^24); BLS12-381 G1 takes c = 17 (255 = 15 x 17) from 2^22 to 2^23; BLS12-381 G2 and everything below 2^20 pairs stay as they were."""
    assert ea.plan(1 << 26, "bls12_377_g1")["window_bits"] == 21
    assert ea.plan(1 << 26, "bls12_377_g2")["window_bits"] == 21
    assert ea.plan(1 << 24, "bls12_377_g1")["window_bits"] == 20
    assert ea.plan(1 << 23, "bls12_377_g1")["window_bits"] == 17
    assert ea.plan(1 << 22, "bls12_381_g1")["window_bits"] == 17
    assert ea.plan(1 << 23, "bls12_381_g1")["window_bits"] == 17
    assert ea.plan(1 << 24, "bls12_381_g1")["window_bits"] == 20
    assert ea.plan(1 << 26, "bls12_381_g1")["window_bits"] == 20
    assert ea.plan(1 << 22, "bls12_381_g2")["window_bits"] == 16
    assert ea.plan((1 << 20) - 1, "bls12_381_g1")["window_bits"] == ea.plan((1 << 20) - 1, "bls12_381_g2")["window_bits"]
