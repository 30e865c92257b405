"""On-disk input formats of the reference harnesses.

* arkworks ``CanonicalSerialize`` files written by the FPGA harness with ``serialize_unchecked``
  (P1B hardcaml/zprize/msm_pippenger/test_fpga_harness/src/util.rs:126-140): ``points.bin`` (``Vec<G1Affine>``),
  ``scalars.bin`` (``Vec<Fr>``), ``arkworks_results.bin`` (``Vec<G1Affine>``).  A ``Vec`` is a little-endian u64 element
  count followed by the elements; an affine point is ``x | y`` as little-endian NORMAL-form integers with ``SWFlags`` in the
  top two bits of the last byte (bit 6 = infinity; P1B nickray driver/algebra/serialize/src/flags.rs:107-134); an ``Fr`` is
  its 32-byte little-endian normal-form integer, which is exactly the ``BigInteger256`` image the MSM ABI takes.
* whitespace-separated hex text (one big-endian hex number per token; points as x then y) as read by
  ``MSMReadHexPoints`` / ``MSMReadHexScalars`` (CMB MSM.cu:77-128) and ``parseHex`` (prize4 yrrid C/Reader.c:10-54).

Point records are handed to the device as they are (``MultiScalarMultContext.set_bases_serialized``); nothing here does
field arithmetic on the host.
"""
from __future__ import annotations

import ctypes
import struct
from typing import List, Tuple

from .msm import MultiScalarMultContext, _check, _curve_id, _COORD_BYTES, load_library, projective_bytes


def record_bytes(curve) -> int:
    """Bytes of one uncompressed serialized affine point (96 for G1, 192 for G2)."""
    return 2 * _COORD_BYTES[_curve_id(curve)]


def read_points_bin(path: str, curve="bls12_377_g1") -> Tuple[bytes, int]:
    """``points.bin`` -> (records without the length prefix, count)."""
    rb = record_bytes(curve)
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        data = f.read(n * rb)
    if len(data) != n * rb:
        raise ValueError(f"{path}: expected {n} records of {rb} bytes, file is short")
    return data, n


def write_points_bin(path: str, records: bytes, curve="bls12_377_g1") -> None:
    rb = record_bytes(curve)
    if len(records) % rb:
        raise ValueError("records length is not a multiple of the record size")
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(records) // rb))
        f.write(records)


def read_scalars_bin(path: str) -> Tuple[bytes, int]:
    """``scalars.bin`` (``Vec<Fr>``) -> (32-byte little-endian integers, count): already the MSM scalar ABI."""
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        data = f.read(n * 32)
    if len(data) != n * 32:
        raise ValueError(f"{path}: expected {n} scalars, file is short")
    return data, n


def write_scalars_bin(path: str, scalars: bytes) -> None:
    if len(scalars) % 32:
        raise ValueError("scalars must be 32-byte integers")
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(scalars) // 32))
        f.write(scalars)


def _hex_tokens(path: str) -> List[str]:
    with open(path) as f:
        return f.read().split()


def read_hex_scalars(path: str, count: int = -1) -> bytes:
    toks = _hex_tokens(path)
    toks = toks if count < 0 else toks[:count]
    return b"".join(int(t, 16).to_bytes(32, "little") for t in toks)


def read_hex_points(path: str, count: int = -1, curve="bls12_377_g1") -> bytes:
    """x y pairs in hex (normal form) -> uncompressed serialized records (G1 only: one token per coordinate)."""
    if _COORD_BYTES[_curve_id(curve)] != 48:
        raise ValueError("hex point files hold G1 points")
    toks = _hex_tokens(path)
    if len(toks) % 2:
        raise ValueError(f"{path}: odd number of coordinates")
    pairs = len(toks) // 2 if count < 0 else count
    out = bytearray()
    for i in range(pairs):
        out += int(toks[2 * i], 16).to_bytes(48, "little") + int(toks[2 * i + 1], 16).to_bytes(48, "little")
    return bytes(out)


def set_bases_serialized(ctx: MultiScalarMultContext, records: bytes) -> None:
    """Upload uncompressed serialized points (see module docstring); conversion to Montgomery form runs on the GPU."""
    rb = record_bytes(ctx.curve)
    if len(records) % rb:
        raise ValueError(f"records must be {rb}-byte uncompressed points")
    n = len(records) // rb
    buf = ctypes.create_string_buffer(records, len(records) or 1)
    lib = load_library()
    _check(lib.mi355_msm_set_bases_serialized(ctx.context, buf, n))
    ctx.npoints = n


def point_to_serialized(projective: bytes, curve="bls12_377_g1") -> bytes:
    """A result (Projective image) as one uncompressed serialized affine record, comparable with arkworks_results.bin."""
    lib = load_library()
    if len(projective) != projective_bytes(curve):
        raise ValueError("wrong projective image size")
    out = ctypes.create_string_buffer(record_bytes(curve))
    buf = ctypes.create_string_buffer(projective, len(projective))
    _check(lib.mi355_msm_point_to_serialized(_curve_id(curve), buf, out))
    return out.raw


def write_results_bin(path: str, results: List[bytes], curve="bls12_377_g1") -> None:
    """``arkworks_results.bin``: the per-batch results as ``Vec<Affine>``."""
    write_points_bin(path, b"".join(point_to_serialized(r, curve) for r in results), curve)
