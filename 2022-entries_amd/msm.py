"""ctypes binding of libmi355msm.so plus the host-side mirror of the reference operator API.

Mirrors (names, argument meaning, error behaviour) the Rust layer every prize1a entry exposes:

  * ``multi_scalar_mult_init(points) -> MultiScalarMultContext`` and
    ``multi_scalar_mult(ctx, points, scalars) -> Vec<G::Projective>`` with
    ``batch_size = scalars.len() / points.len()``
    (P1A 6block/src/lib.rs:54-109, yrrid/src/lib.rs:38-90; the Rust side panics on a non-zero
    error code -- here that is ``MsmError``);
  * ``VariableBaseMSM::msm(bases, scalars)`` truncating to the shorter slice and ``msm_checked``
    (ARK ec/src/msm/variable_base/mod.rs:44-65).

Inputs are the byte images the reference passes across its FFI: arkworks
``G1Affine`` arrays (104-B stride), ``BigInteger256`` arrays (32 B), results are 144-B normalised
``G1Projective`` images.  They may be ``bytes``/``bytearray``/NumPy uint8 arrays (host) or torch uint8
tensors (host or device; device tensors are used in place through their ``data_ptr``).

There is no CPU fallback here: if the HIP library is missing or no GPU is visible, calls raise.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence

CURVE_IDS = {"bls12_377_g1": 0, "bls12_381_g1": 1, "bls12_377_g2": 2, "bls12_381_g2": 3}
AFFINE_STRIDE = 104          # size_of::<G1Affine>()
SCALAR_BYTES = 32
PROJECTIVE_BYTES = 144       # size_of::<G1Projective>()
_COORD_BYTES = {0: 48, 1: 48, 2: 96, 3: 96}   # G2 coordinates live in Fq2 (c0 | c1)


def affine_stride(curve) -> int:
    """size_of::<Affine>() for the curve: two coordinates + the infinity flag, padded to 8 (104 for G1, 200 for G2)."""
    return 2 * _COORD_BYTES[_curve_id(curve)] + 8


def projective_bytes(curve) -> int:
    """size_of::<Projective>(): three coordinates (144 for G1, 288 for G2)."""
    return 3 * _COORD_BYTES[_curve_id(curve)]
T_NAMES = ("digits", "sort", "accumulate", "segreduce", "bucket_reduce", "host_fold", "total")

_LIB = None


class MsmError(RuntimeError):
    """Non-zero RustError from the C ABI (the Rust harness panics here)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"mi355_msm error {code}: {message}")
        self.code = code
        self.message = message


class _RustError(ctypes.Structure):
    _fields_ = [("code", ctypes.c_int), ("message", ctypes.c_void_p)]


def library_path() -> str:
    """The product library; MI355_MSM_LIBRARY names another build of it (tests: libmi355msm_debug.so, the -DMSM_DEBUG build)."""
    override = os.environ.get("MI355_MSM_LIBRARY")
    if override:
        return override if os.path.isabs(override) else os.path.join(os.path.dirname(os.path.abspath(__file__)), override)
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmi355msm.so")


def load_library() -> ctypes.CDLL:
    """Load the HIP shared library built by ``__graft_entry__.build()``; fail loudly when absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build the gfx950 extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the MSM path.")
    # torch bundles its own libamdhip64.so.7; whichever HIP runtime is mapped first serves the whole process.
    # Import torch first so that tensors handed to this library and the library itself share ONE runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    sigs = {
        "mi355_msm_create": [ctypes.POINTER(vp), ci, ci],
        "mi355_msm_create_sharded": [ctypes.POINTER(vp), ci, ctypes.POINTER(ci), ci],
        "mi355_msm_create_env": [ctypes.POINTER(vp), ci],
        "mi355_msm_shard_bounds": [sz, ci, ci, ctypes.POINTER(sz), ctypes.POINTER(sz)],
        "mi355_msm_destroy": [vp],
        "mi355_msm_set_bases": [vp, vp, sz, sz],
        "mi355_msm_set_bases_device": [vp, vp, sz, sz],
        "mi355_msm_run": [vp, vp, vp, sz, sz],
        "mi355_msm_run_device": [vp, vp, vp, sz, sz, vp],
        "mi355_msm_run_async": [vp, vp, vp, sz, sz, vp, vp, vp, ctypes.POINTER(vp)],
        "mi355_msm_job_wait": [vp],
        "mi355_msm_set_option": [vp, ctypes.c_char_p, ctypes.c_long],
        "mi355_msm_last_timings": [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint64)],
        "mi355_msm_query": [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)],
        "mi355_msm_shard_timings": [vp, ci, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint64)],
        "mi355_msm": [ci, vp, vp, sz, vp, sz],
        "mi355_msm_fold": [ci, vp, vp, sz],
        "mi355_msm_generate_points": [ci, ctypes.c_uint64, sz, sz, vp, sz],
        "mi355_msm_plan": [ci, sz, ci, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_uint64)],
        "mi355_msm_set_bases_serialized": [vp, vp, sz],
        "mi355_msm_point_to_serialized": [ci, vp, vp],
        "mi355_msm_last_stateless": [ctypes.POINTER(ctypes.c_double), sz],
        "mi355_msm_stream_create": [ctypes.POINTER(vp), ci, ci, sz, ci],
        "mi355_msm_stream_set_option": [vp, ctypes.c_char_p, ctypes.c_long],
        "mi355_msm_stream_add": [vp, vp, sz, vp, sz],
        "mi355_msm_stream_finalize": [vp, vp],
        "mi355_msm_stream_query": [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)],
        "mi355_msm_stream_destroy": [vp],
        "mi355_msm_trim": [],
        "mi355_msm_pool_stats": [ctypes.POINTER(ctypes.c_uint64), sz],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _RustError
    lib.mi355_msm_version.restype = ctypes.c_char_p
    lib.mi355_msm_job_done.argtypes = [vp]
    lib.mi355_msm_job_done.restype = ci
    _LIB = lib
    return lib


_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]


def _check(err: _RustError) -> None:
    if err.code != 0:
        msg = ctypes.string_at(err.message).decode("utf-8", "replace") if err.message else "(no message)"
        if err.message:
            _libc.free(err.message)
        raise MsmError(err.code, msg)


def _curve_id(curve) -> int:
    if isinstance(curve, int):
        return curve
    try:
        return CURVE_IDS[curve]
    except KeyError:
        raise ValueError(f"unknown curve {curve!r}; known: {sorted(CURVE_IDS)}") from None


class _Buf:
    """Uniform view of bytes / numpy / torch inputs: pointer, byte length, device flag, keep-alive."""

    def __init__(self, obj):
        self.keep = obj
        self.is_device = False
        self.stream = None
        if hasattr(obj, "data_ptr") and hasattr(obj, "is_cuda"):  # torch tensor
            import torch

            t = obj
            if t.dtype != torch.uint8:
                raise TypeError("tensor inputs must be torch.uint8 byte images")
            if not t.is_contiguous():
                t = t.contiguous()
            self.keep = t
            self.ptr = t.data_ptr()
            self.nbytes = t.numel()
            self.is_device = bool(t.is_cuda)
            if self.is_device:
                self.device_index = t.device.index if t.device.index is not None else torch.cuda.current_device()
                self.stream = torch.cuda.current_stream(t.device).cuda_stream
        elif hasattr(obj, "__array_interface__"):  # numpy
            import numpy as np

            a = np.ascontiguousarray(obj)
            if a.dtype != np.uint8:
                a = a.view(np.uint8)
            self.keep = a
            self.ptr = a.ctypes.data
            self.nbytes = a.nbytes
        else:
            b = obj if isinstance(obj, (bytes, bytearray)) else bytes(obj)
            self.keep = (ctypes.c_char * len(b)).from_buffer_copy(b) if len(b) else ctypes.create_string_buffer(1)
            self.ptr = ctypes.addressof(self.keep)
            self.nbytes = len(b)


def _flat_bytes(obj):
    """1-D byte view of a bytes / numpy / torch input, sliceable by byte offset."""
    if hasattr(obj, "data_ptr") and hasattr(obj, "is_cuda"):
        return obj.contiguous().reshape(-1)
    if hasattr(obj, "__array_interface__"):
        import numpy as np

        return np.ascontiguousarray(obj).view(np.uint8).reshape(-1)
    return memoryview(obj if isinstance(obj, (bytes, bytearray)) else bytes(obj))


class MultiScalarMultContext:
    """``#[repr(C)] struct MultiScalarMultContext { context: *mut c_void }`` (P1A 6block/src/lib.rs:18-21)."""

    def __init__(self, curve="bls12_377_g1", device: Optional[int] = None, devices: Optional[Sequence[int]] = None):
        """``device``: one GPU (None = the current HIP device).  ``devices``: a SHARDED context over those GPUs, one slice of the
        bases and of every scalar batch per entry (an id may repeat: logical shards on one GPU); same methods, same results."""
        self.curve = _curve_id(curve)
        self._lib = load_library()
        self.context = ctypes.c_void_p()
        self.devices = None if devices is None else [int(d) for d in devices]
        if self.devices is not None:
            arr = (ctypes.c_int * len(self.devices))(*self.devices)
            _check(self._lib.mi355_msm_create_sharded(ctypes.byref(self.context), self.curve, arr, len(self.devices)))
            self.device = None
        else:
            _check(self._lib.mi355_msm_create(ctypes.byref(self.context), self.curve, -1 if device is None else device))
            self.device = self.query("device")
        self.npoints = 0

    def _check_device(self, b: "_Buf", what: str) -> None:
        # a pointer from another GPU would be dereferenced after hipSetDevice(ctx.device): fail with a clear message instead
        if b.is_device and self.device is not None and b.device_index != self.device:
            raise MsmError(-1, f"{what} live on cuda:{b.device_index} but this context is bound to device {self.device}")

    @classmethod
    def from_env(cls, curve="bls12_377_g1") -> "MultiScalarMultContext":
        """What the harness shims do: honour MI355_MSM_DEVICES ("0,1,2,3", "0-7", "all"; unset = the current device),
        MI355_MSM_ASSUME_SUBGROUP (0 | 1: the context option of that name) and MI355_MSM_PRECOMPUTE (auto | 1 | 0) /
        MI355_MSM_TABLE_LEVELS (options "precompute" / "table_levels")."""
        self = cls.__new__(cls)
        self.curve = _curve_id(curve)
        self._lib = load_library()
        self.context = ctypes.c_void_p()
        _check(self._lib.mi355_msm_create_env(ctypes.byref(self.context), self.curve))
        self.npoints = 0
        self.devices = None
        self.device = None if self.query("shards") else self.query("device")
        return self

    def set_bases(self, points, stride: Optional[int] = None) -> None:
        stride = affine_stride(self.curve) if stride is None else stride
        b = _Buf(points)
        if b.nbytes % stride:
            raise ValueError(f"points image of {b.nbytes} bytes is not a multiple of the {stride}-byte affine stride")
        n = b.nbytes // stride
        self._check_device(b, "bases")
        fn = self._lib.mi355_msm_set_bases_device if b.is_device else self._lib.mi355_msm_set_bases
        _check(fn(self.context, b.ptr, n, stride))
        self.npoints = n

    def run(self, scalars, npoints: Optional[int] = None) -> List[bytes]:
        b = _Buf(scalars)
        self._check_device(b, "scalars")
        n = self.npoints if npoints is None else npoints
        if b.nbytes % SCALAR_BYTES:
            raise ValueError("scalars image is not a multiple of 32 bytes")
        count = b.nbytes // SCALAR_BYTES
        if n == 0:
            batches = 1 if count == 0 else None
        else:
            batches = count // n if count % n == 0 else None
        if batches is None:
            raise ValueError(f"{count} scalars is not a whole number of batches of {n} points")
        pb = projective_bytes(self.curve)
        out = ctypes.create_string_buffer(pb * max(batches, 1))
        if b.is_device:
            _check(self._lib.mi355_msm_run_device(self.context, out, b.ptr, n, batches, b.stream))
        else:
            _check(self._lib.mi355_msm_run(self.context, out, b.ptr, n, batches))
        raw = out.raw
        return [raw[i * pb:(i + 1) * pb] for i in range(batches)]

    def run_async(self, scalars, npoints: Optional[int] = None, stream=None) -> "MsmJob":
        """Stream-ordered run (mi355_msm_run_async; the role of ML bellman-cuda.h:48-75 msm_execute_async): returns at once with a job
        handle.  ``scalars`` must be a device tensor; the MSM is ordered after the work already enqueued in ``stream`` (default: the
        tensor's current torch stream) and runs on the context's own stream, so whatever the caller launches next overlaps it.
        ``job.wait()`` returns the projective images; jobs of one context run in submission order."""
        b = _Buf(scalars)
        if not b.is_device:
            raise TypeError("run_async takes scalars that are resident on the device (use run() for host scalars)")
        self._check_device(b, "scalars")
        n = self.npoints if npoints is None else npoints
        count = b.nbytes // SCALAR_BYTES
        if b.nbytes % SCALAR_BYTES or n == 0 or count % n:
            raise ValueError(f"{count} scalars is not a whole number of batches of {n} points")
        batches = count // n
        pb = projective_bytes(self.curve)
        out = ctypes.create_string_buffer(pb * batches)
        job = ctypes.c_void_p()
        st = b.stream if stream is None else (stream.cuda_stream if hasattr(stream, "cuda_stream") else stream)
        _check(self._lib.mi355_msm_run_async(self.context, out, b.ptr, n, batches, st, None, None, ctypes.byref(job)))
        return MsmJob(self._lib, job, out, pb, batches, b)

    def set_option(self, key: str, value: int) -> None:
        _check(self._lib.mi355_msm_set_option(self.context, key.encode(), int(value)))

    def query(self, key: str) -> int:
        """Context state: "twisted_edwards", "twisted_edwards_fallbacks", "twisted_edwards_demotions", "oom_backoffs", "chunk_cap",
        "device", "shards", "rccl_exchanges", "peer_stagings", "bases", "table_levels", "table_window_bits", "base_bytes", "assume_subgroup",
        "carry", "precompute", "g2_paired", "anchor", "anchored_window", "anchor_sums", "anchor_sum_us", and the geometry of the most recent chunk: "bucket_windows", "l1_bits", "l1_bins", "group_passes"."""
        v = ctypes.c_uint64(0)
        _check(self._lib.mi355_msm_query(self.context, key.encode(), ctypes.byref(v)))
        return int(v.value)

    def last_timings(self) -> dict:
        ms = (ctypes.c_float * 8)()
        info = (ctypes.c_uint64 * 8)()
        _check(self._lib.mi355_msm_last_timings(self.context, ms, info))
        d = {name: float(ms[i]) for i, name in enumerate(T_NAMES)}
        d.update(window_bits=int(info[0]), windows=int(info[1]), entries=int(info[2]), lane_entries=int(info[3]),
                 launches=int(info[4]), lanes=int(info[5]), tables=bool(info[6]), twisted_edwards=bool(info[7]))
        return d

    def shard_timings(self) -> list:
        """Per-shard stage times of the most recent run (one entry for an unsharded context)."""
        out = []
        for g in range(max(1, self.query("shards"))):
            ms = (ctypes.c_float * 8)()
            info = (ctypes.c_uint64 * 8)()
            _check(self._lib.mi355_msm_shard_timings(self.context, g, ms, info))
            d = {name: float(ms[i]) for i, name in enumerate(T_NAMES)}
            d.update(shard=g, window_bits=int(info[0]), entries=int(info[2]), launches=int(info[4]))
            out.append(d)
        return out

    def close(self) -> None:
        if self.context:
            _check(self._lib.mi355_msm_destroy(self.context))
            self.context = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def multi_scalar_mult_init(points, curve="bls12_377_g1", device: Optional[int] = None,
                           devices: Optional[Sequence[int]] = None) -> MultiScalarMultContext:
    """Upload (and convert) the fixed base vector once; untimed in the reference bench (benches/msm.rs:21).
    ``devices`` shards the vector over several GPUs behind the same context (see MultiScalarMultContext)."""
    if devices is None and device is None and os.environ.get("MI355_MSM_DEVICES"):
        ctx = MultiScalarMultContext.from_env(curve)
        ctx.set_bases(points)
        return ctx
    ctx = MultiScalarMultContext(curve, device, devices)
    ctx.set_bases(points)
    return ctx


def multi_scalar_mult(ctx: MultiScalarMultContext, points, scalars) -> List[bytes]:
    """One 144-byte projective image per batch; ``points`` is only used for its length, as in the reference
    (``npoints = points.len()``, P1A 6block/src/lib.rs:92-101)."""
    npoints = ctx.npoints if points is None else _Buf(points).nbytes // affine_stride(ctx.curve)
    if npoints != ctx.npoints:
        raise MsmError(-1, f"context was initialised with {ctx.npoints} points, called with {npoints}")
    return ctx.run(scalars, npoints)


def msm(bases, scalars, curve="bls12_377_g1") -> bytes:
    """Stateless ``msm(bases, scalars, n)``; chops to the shorter input like VariableBaseMSM::msm."""
    stride = affine_stride(curve)
    nb = _Buf(bases).nbytes // stride
    ns = _Buf(scalars).nbytes // SCALAR_BYTES
    n = min(nb, ns)
    pb, sb = _Buf(bases), _Buf(scalars)
    if not (pb.is_device or sb.is_device):
        # both operands in host memory: the stateless C entry (a pipeline: slices cross PCIe while earlier ones compute)
        out = ctypes.create_string_buffer(projective_bytes(curve))
        _check(load_library().mi355_msm(_curve_id(curve), out, pb.ptr, n, sb.ptr, stride))
        return out.raw
    ctx = MultiScalarMultContext(curve)
    try:
        # chop BYTES, not rows: the inputs are usually 2-D (N, 104) / (N, 32) tensors
        ctx.set_bases(_flat_bytes(bases)[: n * stride])
        return ctx.run(_flat_bytes(scalars)[: n * SCALAR_BYTES], n)[0]
    finally:
        ctx.close()


def last_stateless() -> dict:
    """What this thread's most recent stateless ``msm`` call did (mi355_msm_last_stateless)."""
    v = (ctypes.c_double * 10)()
    _check(load_library().mi355_msm_last_stateless(v, 10))
    names = ("total_ms", "setup_ms", "wait_upload_ms", "compute_ms", "tail_ms", "slices", "threads", "bytes", "dma_done_ms", "first_dma_ms")
    return {k: float(v[i]) for i, k in enumerate(names)}


def trim() -> None:
    """Give back what the stateless entry points keep between calls (pinned rings, idle contexts): mi355_msm_trim."""
    _check(load_library().mi355_msm_trim())


def pool_stats() -> dict:
    """mi355_msm_pool_stats: idle contexts of the stateless path and the memory they hold."""
    v = (ctypes.c_uint64 * 4)()
    _check(load_library().mi355_msm_pool_stats(v, 4))
    return {"idle_contexts": int(v[0]), "idle_device_bytes": int(v[1]), "idle_rings": int(v[2]), "idle_pinned_bytes": int(v[3])}


class VariableBaseMSM:
    """Shape of the arkworks trait (ARK ec/src/msm/variable_base/mod.rs:15-65) for one curve."""

    def __init__(self, curve="bls12_377_g1"):
        self.curve = curve

    def msm(self, bases, scalars) -> bytes:
        return msm(bases, scalars, self.curve)

    def msm_checked(self, bases, scalars):
        """``Ok(point)`` as bytes, or ``Err(min_len)`` as an int when lengths differ."""
        nb = _Buf(bases).nbytes // affine_stride(self.curve)
        ns = _Buf(scalars).nbytes // SCALAR_BYTES
        if nb != ns:
            return min(nb, ns)
        return self.msm(bases, scalars)

    msm_bigint = msm

    def msm_chunks(self, bases_stream, scalars_stream, step: int = 1 << 20) -> bytes:
        """Streaming form (ARK ec/src/msm/variable_base/mod.rs:165-199): ``scalars_stream`` holds ``Fr`` values (Montgomery
        form, converted on the device like ``into_bigint``) and must not be longer than ``bases_stream``; the LAST
        ``len(scalars)`` bases are used ("align the streams"), ``step`` pairs at a time, and the partial sums are added.
        The reference hard-codes step = 2^20; any step gives the same point."""
        stride = affine_stride(self.curve)
        bases_stream, scalars_stream = _flat_bytes(bases_stream), _flat_bytes(scalars_stream)
        nb, ns = len(bases_stream) // stride, len(scalars_stream) // SCALAR_BYTES
        if ns > nb:
            raise MsmError(-1, f"msm_chunks: {ns} scalars for {nb} bases (scalars_stream.len() <= bases_stream.len())")
        if step <= 0:
            raise MsmError(-1, "msm_chunks: step must be positive")
        skip = nb - ns
        ctx = MultiScalarMultContext(self.curve)
        try:
            ctx.set_option("scalars_montgomery", 1)
            partials = []
            for lo in range(0, ns, step):
                hi = min(ns, lo + step)
                ctx.set_bases(bases_stream[(skip + lo) * stride:(skip + hi) * stride])
                partials.append(ctx.run(scalars_stream[lo * SCALAR_BYTES:hi * SCALAR_BYTES], hi - lo)[0])
            return fold_partials(partials, self.curve)
        finally:
            ctx.close()


class ChunkedPippenger:
    """``ChunkedPippenger::{new, with_size, add, finalize}`` (ARK ec/src/msm/variable_base/stream_pippenger.rs:11-75): buffer
    (base, BigInt scalar) pairs; every ``buf_size`` pairs ``result += msm_bigint(buffer)``; ``finalize`` flushes the rest.
    ``add`` takes one pair or arrays of pairs (byte images as everywhere in this module)."""

    _HASHMAP = 0

    def __init__(self, max_msm_buffer: int, curve="bls12_377_g1", device: Optional[int] = None):
        self.curve = _curve_id(curve)
        self._lib = load_library()
        self.stream = ctypes.c_void_p()
        _check(self._lib.mi355_msm_stream_create(ctypes.byref(self.stream), self.curve, -1 if device is None else device,
                                                 max_msm_buffer, self._HASHMAP))

    new = classmethod(lambda cls, max_msm_buffer, curve="bls12_377_g1": cls(max_msm_buffer, curve))
    with_size = new

    def set_option(self, key: str, value: int) -> None:
        _check(self._lib.mi355_msm_stream_set_option(self.stream, key.encode(), int(value)))

    def add(self, bases, scalars, stride: Optional[int] = None) -> None:
        stride = affine_stride(self.curve) if stride is None else stride
        b, s = _Buf(bases), _Buf(scalars)
        if b.is_device or s.is_device:
            raise TypeError("the streaming accumulators buffer pairs on the host: pass host arrays")
        if b.nbytes % stride or s.nbytes % SCALAR_BYTES or b.nbytes // stride != s.nbytes // SCALAR_BYTES:
            raise ValueError("bases and scalars must hold the same number of pairs")
        _check(self._lib.mi355_msm_stream_add(self.stream, b.ptr, stride, s.ptr, s.nbytes // SCALAR_BYTES))

    def finalize(self) -> bytes:
        out = ctypes.create_string_buffer(projective_bytes(self.curve))
        _check(self._lib.mi355_msm_stream_finalize(self.stream, out))
        return out.raw

    def query(self, key: str) -> int:
        v = ctypes.c_uint64(0)
        _check(self._lib.mi355_msm_stream_query(self.stream, key.encode(), ctypes.byref(v)))
        return int(v.value)

    def close(self) -> None:
        if self.stream:
            _check(self._lib.mi355_msm_stream_destroy(self.stream))
            self.stream = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HashMapPippenger(ChunkedPippenger):
    """``HashMapPippenger::{new, add, finalize}`` (stream_pippenger.rs:78-140): a pair whose base is already buffered adds its
    scalar (an ``Fr`` value) to that entry modulo r; the MSM runs when ``max_msm_buffer`` DISTINCT bases are buffered."""

    _HASHMAP = 1


def fold_partials(partials: Sequence[bytes], curve="bls12_377_g1") -> bytes:
    """Sum per-GPU partial results (projective images: 144 B for G1, 288 B for G2) into one normalised image."""
    lib = load_library()
    pb = projective_bytes(curve)
    blob = b"".join(bytes(p) for p in partials)
    if len(blob) % pb:
        raise ValueError(f"partials must be {pb}-byte projective images")
    out = ctypes.create_string_buffer(pb)
    buf = ctypes.create_string_buffer(blob, len(blob) if blob else 1)
    _check(lib.mi355_msm_fold(_curve_id(curve), out, buf, len(blob) // pb))
    return out.raw


def generate_points(npoints: int, distinct: int = 1 << 15, seed: int = 0x5A5052495A45, curve="bls12_377_g1"):
    """Synthetic bases in the reference generator's shape (P1A yrrid/src/util.rs:15-28): ``distinct`` subgroup points
    replicated by doubling up to ``npoints``; returns a NumPy uint8 array of shape (npoints, stride) (stride 104, G2: 200)."""
    import numpy as np

    lib = load_library()
    stride = affine_stride(curve)
    out = np.zeros((npoints, stride), dtype=np.uint8)
    _check(lib.mi355_msm_generate_points(_curve_id(curve), seed, distinct, npoints, out.ctypes.data, stride))
    return out


class MsmJob:
    """A run in flight (MultiScalarMultContext.run_async).  Keeps the scalars and the output buffer alive until it is waited for."""

    def __init__(self, lib, handle, out, pb, batches, keep):
        self._lib, self._handle, self._out, self._pb, self._batches, self._keep = lib, handle, out, pb, batches, keep
        self._result = None

    def done(self) -> bool:
        return self._handle is None or bool(self._lib.mi355_msm_job_done(self._handle))

    def wait(self) -> List[bytes]:
        if self._handle is not None:
            h, self._handle = self._handle, None
            _check(self._lib.mi355_msm_job_wait(h))      # (releases the handle, whatever the status)
            raw = self._out.raw
            self._result = [raw[i * self._pb:(i + 1) * self._pb] for i in range(self._batches)]
            self._keep = None
        if self._result is None:
            raise MsmError(-1, "this job failed (its error was raised by the first wait())")
        return self._result

    def __del__(self):
        try:
            if self._handle is not None:
                self.wait()
        except Exception:
            pass


def plan(npoints: int, curve="bls12_377_g1", precompute: bool = False, window_bits: int = 0, lane_entries: int = 0,
         seg_entries: int = 0, table_levels: int = 0) -> dict:
    """The engine's execution plan for an MSM of ``npoints`` pairs (host arithmetic only; works without a GPU).
    ``precompute`` with ``table_levels`` = k > 1 plans what a context with those two options runs (k levels, ceil(windows / k)
    bucket sets); ``table_levels`` = 0 is a level per window."""
    lib = load_library()
    opts = (ctypes.c_long * 3)(window_bits, lane_entries, seg_entries)
    out = (ctypes.c_uint64 * 10)()
    if table_levels == 1:
        raise ValueError("table_levels = 1 is no table at all: pass precompute=False")
    _check(lib.mi355_msm_plan(_curve_id(curve), npoints, (table_levels if table_levels > 1 else 1) if precompute else 0, opts, out))
    names = ("window_bits", "windows", "bucket_windows", "entries", "lane_entries", "lanes", "merge_launches", "reduce_launches",
             "key_bits", "work_bytes")
    return {k: int(out[i]) for i, k in enumerate(names)}
