"""CPU: MSMReadHexPoints / MSMReadHexScalars as C symbols of the yrrid shim (CMB MSM.h:68-69, MSM.cu:82-128; token grammar of
parseHex, prize4 yrrid C/Reader.c:10-54) against the Python readers of 2022-entries_amd/formats.py on the same files."""
import ctypes
import os
import random

import numpy as np
import pytest


@pytest.fixture(scope="module")
def shim(ea):
    lib = ctypes.CDLL(os.path.join(ea.PACKAGE_DIR, "libmi355msm_yrrid_377.so"))
    lib.MSMReadHexPoints.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p]
    lib.MSMReadHexScalars.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p]
    return lib


def test_hex_scalars_and_points_match_the_python_readers(ea, shim, tmp_path):
    rng = random.Random(4)
    n = 257
    scalars = [0, 1, (1 << 256) - 1, 0xABCDEF] + [rng.randrange(1 << 253) for _ in range(n - 4)]
    coords = [(rng.randrange(1 << 377), rng.randrange(1 << 377)) for _ in range(n)]
    coords[0] = (0, 1)
    sp, pp = tmp_path / "scalars.hex", tmp_path / "points.hex"
    # mixed case, short tokens (zero-extended), blank lines, tabs and CRLF: whatever separates tokens is white space
    sp.write_text("\r\n\n".join(("%x" % s).upper() if i % 3 == 0 else ("%064x" % s) for i, s in enumerate(scalars)) + "\n")
    pp.write_text("\n".join("%x\t%096X" % xy if i % 2 else "  %096x \r\n %x" % xy for i, xy in enumerate(coords)))
    sc = np.full((n, 32), 0xEE, dtype=np.uint8)
    assert shim.MSMReadHexScalars(sc.ctypes.data, n, str(sp).encode()) == 0
    assert sc.tobytes() == ea.formats.read_hex_scalars(str(sp)) == b"".join(s.to_bytes(32, "little") for s in scalars)
    pts = np.full((n, 104), 0xEE, dtype=np.uint8)
    assert shim.MSMReadHexPoints(pts.ctypes.data, n, str(pp).encode()) == 0
    rec = ea.formats.read_hex_points(str(pp))
    for i in range(n):
        assert pts[i, :96].tobytes() == rec[96 * i:96 * (i + 1)] == coords[i][0].to_bytes(48, "little") + coords[i][1].to_bytes(48, "little")
        assert pts[i, 96:].tobytes() == bytes(8)          # the flag word is cleared (CMB MSM.cu:98-99)
    # a prefix of the file is fine; more than the file holds, a bad digit, an over-long token and a missing file are errors
    assert shim.MSMReadHexScalars(sc.ctypes.data, 10, str(sp).encode()) == 0
    assert shim.MSMReadHexScalars(np.zeros((n + 1, 32), dtype=np.uint8).ctypes.data, n + 1, str(sp).encode()) == -1
    bad = tmp_path / "bad.hex"
    bad.write_text("12 3g 45")
    assert shim.MSMReadHexScalars(sc.ctypes.data, 3, str(bad).encode()) == -1
    bad.write_text("1" * 65)
    assert shim.MSMReadHexScalars(sc.ctypes.data, 1, str(bad).encode()) == -1
    assert shim.MSMReadHexPoints(pts.ctypes.data, 1, str(tmp_path / "nope").encode()) == -1
