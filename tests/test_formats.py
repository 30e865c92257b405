"""Row f2: the reference harnesses' on-disk formats -- arkworks CanonicalSerialize files (points.bin / scalars.bin /
arkworks_results.bin, P1B hardcaml/.../test_fpga_harness/src/util.rs:126-140) and hex text
(CMB MSM.cu:77-128, prize4 yrrid C/Reader.c:10-54)."""
import os
import random
import struct

import numpy as np
import pytest

import pymodel as m
from conftest import oracle_msm


def _case(curve, n, seed):
    rng = random.Random(seed)
    pts = m.random_points(curve, n, rng, max(1, n // 4))
    if n > 5:
        pts[3] = None
    sc = m.random_scalars(curve, n, rng)
    return pts, sc


def test_file_layouts_roundtrip(ea, tmp_path):
    c = m.BLS12_377_G1
    pts, sc = _case(c, 9, 1)
    records = b"".join(c.encode_serialized(P) for P in pts)
    f = ea.formats
    f.write_points_bin(str(tmp_path / "points.bin"), records)
    f.write_scalars_bin(str(tmp_path / "scalars.bin"), m.encode_scalars(sc))
    raw = open(tmp_path / "points.bin", "rb").read()
    assert struct.unpack("<Q", raw[:8])[0] == 9 and len(raw) == 8 + 9 * 96
    assert raw[8 + 3 * 96 + 95] == 0x40                       # SWFlags::Infinity on the planted infinity point
    assert f.read_points_bin(str(tmp_path / "points.bin")) == (records, 9)
    assert f.read_scalars_bin(str(tmp_path / "scalars.bin")) == (m.encode_scalars(sc), 9)
    with open(tmp_path / "points.hex", "w") as h:
        for P in pts[:3]:
            h.write("%X\n%x\n" % P)
    with open(tmp_path / "scalars.hex", "w") as h:
        h.write(" ".join("%x" % k for k in sc))
    assert f.read_hex_points(str(tmp_path / "points.hex")) == records[:3 * 96]
    assert f.read_hex_scalars(str(tmp_path / "scalars.hex"), 4) == m.encode_scalars(sc[:4])
    with pytest.raises(ValueError):
        f.write_points_bin(str(tmp_path / "bad.bin"), records[:100])


@pytest.mark.parametrize("curve", [m.BLS12_377_G1, m.BLS12_381_G1, m.BLS12_377_G2])
def test_result_serialization_is_host_side(ea, curve):
    """mi355_msm_point_to_serialized: Projective image (any Z) -> uncompressed CanonicalSerialize record."""
    rng = random.Random(4)
    pts = m.random_points(curve, 3, rng)
    for P in pts + [None]:
        assert ea.formats.point_to_serialized(curve.encode_projective_normalized(P), curve.name) == curve.encode_serialized(P)


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [m.BLS12_377_G1, m.BLS12_381_G1, m.BLS12_377_G2])
def test_msm_from_serialized_files(ea, oracle, tmp_path, curve):
    import ctypes

    n = 700
    pts, sc = _case(curve, n, 9)
    f = ea.formats
    f.write_points_bin(str(tmp_path / "points.bin"), b"".join(curve.encode_serialized(P) for P in pts), curve.name)
    f.write_scalars_bin(str(tmp_path / "scalars.bin"), m.encode_scalars(sc))
    records, np_ = f.read_points_bin(str(tmp_path / "points.bin"), curve.name)
    scalars, ns = f.read_scalars_bin(str(tmp_path / "scalars.bin"))
    assert np_ == ns == n
    ctx = ea.MultiScalarMultContext(curve.name)
    f.set_bases_serialized(ctx, records)
    got = ctx.run(scalars)[0]
    bases = curve.encode_affine_array(pts)
    out = ctypes.create_string_buffer(curve.projective_bytes)
    assert oracle.oracle_msm(curve.curve_id, ctypes.create_string_buffer(bases, len(bases)), curve.affine_stride,
                             ctypes.create_string_buffer(scalars, len(scalars)), n, out, 0) == 0
    assert got == out.raw
    f.write_results_bin(str(tmp_path / "arkworks_results.bin"), [got], curve.name)
    rec, cnt = f.read_points_bin(str(tmp_path / "arkworks_results.bin"), curve.name)
    assert cnt == 1 and rec == curve.encode_serialized(curve.decode_projective(got))
    ctx.close()
