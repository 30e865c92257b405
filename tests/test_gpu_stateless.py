"""GPU: the stateless call msm(bases, scalars, n) -- mi355_msm() and sppark's 5-argument mult_pippenger_inf -- is a pipeline
(csrc/msm_stateless.hpp): slices of both operands are staged through a pinned ring by host threads and cross PCIe while
earlier slices are converted and run.  Same bytes as the oracle and as the context path for every slicing, thread count,
curve and edge case; the caller's current device is left alone.

Reference behaviour: SPK poc/blst-cuda/cuda/pippenger_inf.cu:28-35 (the entry), SPK msm/pippenger.cuh:617-661 (upload of the next
slice overlapped with the current one), P1A matter-labs/src/lib.rs:171-182 (growing chunks)."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT, oracle_msm_np

pytestmark = pytest.mark.gpu

R377_TOP = 0x12ab655e9a2ca556


def _scalars(n, seed, top=R377_TOP):
    rng = np.random.default_rng(seed)
    limbs = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] %= np.uint64(top)
    return limbs.view(np.uint8).reshape(n, 32)


class _Env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("curve,cid", [("bls12_377_g1", 0), ("bls12_381_g1", 1), ("bls12_377_g2", 2), ("bls12_381_g2", 3)])
def test_stateless_matches_oracle_for_every_slicing(ea, oracle, curve, cid):
    """Slices of 2^10 / 2^12 pairs (ragged last slice, more slices than raw-record buffers, so the ring of three wraps),
    1 and 5 staging threads, and the automatic choice."""
    stride = ea.affine_stride(curve)
    for n in (1, 2, 777, 4096 + 1, 20000 if cid < 2 else 6000):
        bases = ea.generate_points(n, distinct=max(1, n // 3), seed=n, curve=curve)
        sc = _scalars(n, 11 * n + cid)
        if n > 10:
            sc[3] = 0
            sc[4] = 0
            sc[4, 0] = 1
            bases[7, stride - 8] = 1          # a base at infinity (the flag byte is authoritative)
        exp = ctypes.create_string_buffer(ea.projective_bytes(curve))
        assert oracle.oracle_msm(cid, bases.ctypes.data, stride, sc.ctypes.data, n, exp, 0) == 0
        for env in ({}, {"MI355_MSM_STATELESS_SLICE_LOG": 10, "MI355_MSM_STAGE_THREADS": 5},
                    {"MI355_MSM_STATELESS_SLICE_LOG": 12, "MI355_MSM_STAGE_THREADS": 1}):
            with _Env(**env):
                assert ea.msm(bases, sc, curve) == exp.raw, (curve, n, env)
                st = ea.last_stateless()
                if env:
                    slice_pairs = 1 << env["MI355_MSM_STATELESS_SLICE_LOG"]
                    assert st["slices"] >= max(1, n // slice_pairs - 1) and st["threads"] <= env["MI355_MSM_STAGE_THREADS"]
                assert st["bytes"] == n * (stride + 32)


def test_stateless_empty_and_argument_errors(ea):
    lib = ea.load_library()
    out = ctypes.create_string_buffer(144)
    e = lib.mi355_msm(0, out, None, 0, None, 104)
    assert e.code == 0 and out.raw[96:] == bytes(48)      # (1, 1, 0): the point at infinity
    e = lib.mi355_msm(0, out, None, 5, None, 104)
    assert e.code != 0 and b"null" in ctypes.string_at(e.message)
    ctypes.CDLL(None).free(ctypes.c_void_p(e.message))
    bases = ea.generate_points(4, distinct=4, seed=1)
    sc = _scalars(4, 1)
    e = lib.mi355_msm(0, out, bases.ctypes.data, 4, sc.ctypes.data, 50)
    assert e.code != 0 and b"stride" in ctypes.string_at(e.message)
    ctypes.CDLL(None).free(ctypes.c_void_p(e.message))
    e = lib.mi355_msm(0, None, bases.ctypes.data, 4, sc.ctypes.data, 104)
    assert e.code != 0
    ctypes.CDLL(None).free(ctypes.c_void_p(e.message))


def test_stateless_wide_stride_and_unaligned_views(ea, oracle):
    """ffi_affine_sz is the caller's element stride (sppark passes sizeof(affine_t)); a 112-byte stride and operands that do
    not start on a page."""
    n = 3001
    bases = ea.generate_points(n, distinct=500, seed=5)
    wide = np.zeros((n, 112), dtype=np.uint8)
    wide[:, :104] = bases
    backing = np.zeros(n * 32 + 24, dtype=np.uint8)
    sc = backing[24:].reshape(n, 32)
    sc[:] = _scalars(n, 6)
    out = ctypes.create_string_buffer(144)
    e = ea.load_library().mi355_msm(0, out, wide.ctypes.data, n, sc.ctypes.data, 112)
    assert e.code == 0
    assert out.raw == oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(sc), n)


def test_sppark_shim_2_24_equals_context_path(ea, oracle):
    """The north-star entry point at a size where the pipeline has five slices: libmi355msm_sppark_377.so's 5-argument
    mult_pippenger_inf == the context path (twisted-Edwards law, one chunk) == the oracle on a prefix."""
    import torch

    class RustError(ctypes.Structure):
        _fields_ = [("code", ctypes.c_int), ("message", ctypes.c_char_p)]

    n = 1 << 24
    tile = ea.generate_points(1 << 13, distinct=1 << 13, seed=99)
    bases = np.ascontiguousarray(np.tile(tile, (n >> 13, 1)))
    sc = _scalars(n, 1234)
    shim = ctypes.CDLL(os.path.join(ea.PACKAGE_DIR, "libmi355msm_sppark_377.so"))
    shim.mult_pippenger_inf.restype = RustError
    shim.mult_pippenger_inf.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    out = ctypes.create_string_buffer(144)
    e = shim.mult_pippenger_inf(out, bases.ctypes.data, n, sc.ctypes.data, 104)
    assert e.code == 0, e.message
    st = ea.last_stateless()
    assert st["slices"] >= 4 and st["bytes"] == n * 136
    ctx = ea.multi_scalar_mult_init(torch.from_numpy(bases).cuda(), "bls12_377_g1")
    assert ctx.run(sc)[0] == out.raw
    assert ctx.query("twisted_edwards") == 1
    ctx.close()
    # a second call reuses the staging ring; a prefix small enough for the oracle
    k = 1 << 16
    e = shim.mult_pippenger_inf(out, bases.ctypes.data, k, sc.ctypes.data, 104)
    assert e.code == 0, e.message
    assert out.raw == oracle_msm_np(oracle, 0, bases, sc, k)
    assert ea.load_library().mi355_msm_trim().code == 0


def test_stateless_sharded_by_environment(ea, oracle):
    """MI355_MSM_DEVICES names several (here: logical) shards: one pipeline per shard over its slice of both operands."""
    n = 30011
    bases = ea.generate_points(n, distinct=999, seed=31)
    sc = _scalars(n, 32)
    exp = oracle_msm_np(oracle, 0, bases, sc, n)
    with _Env(MI355_MSM_DEVICES="0,0,0", MI355_MSM_STATELESS_SLICE_LOG=12):
        assert ea.msm(bases, sc) == exp
    with _Env(MI355_MSM_DEVICES="0,0,0,0,0,0,0,0"):
        assert ea.msm(bases[:5], sc[:5]) == oracle_msm_np(oracle, 0, bases, sc, 5)      # more shards than pairs


def test_calls_leave_the_current_device_alone(ea, oracle):
    """A context bound to (or sharded over) other GPUs must not retarget the calling thread: torch.cuda.current_device() and
    hipGetDevice are the same before and after set_bases / run / the stateless call / destroy."""
    import torch

    ndev = torch.cuda.device_count()
    n = 5000
    bases = ea.generate_points(n, distinct=500, seed=3)
    sc = _scalars(n, 3)
    exp = oracle_msm_np(oracle, 0, bases, sc, n)
    torch.cuda.set_device(0)
    others = list(range(ndev))[::-1]            # last device first: on a multi-GPU box shard 0 is NOT the current device
    ctx = ea.MultiScalarMultContext("bls12_377_g1", devices=others if ndev > 1 else [0, 0])
    ctx.set_bases(bases)
    assert torch.cuda.current_device() == 0
    assert ctx.run(sc)[0] == exp
    assert torch.cuda.current_device() == 0
    assert ctx.run(torch.from_numpy(sc).cuda())[0] == exp
    assert torch.cuda.current_device() == 0
    ctx.close()
    assert torch.cuda.current_device() == 0
    if ndev > 1:
        c1 = ea.MultiScalarMultContext("bls12_377_g1", device=ndev - 1)
        c1.set_bases(bases)
        assert c1.run(sc)[0] == exp and torch.cuda.current_device() == 0
        c1.close()
        with _Env(MI355_MSM_DEVICES="all"):
            assert ea.msm(bases, sc) == exp
        assert torch.cuda.current_device() == 0
    # a tensor allocated now lands on device 0
    assert torch.zeros(1, device="cuda").device.index == 0


def test_concurrent_stateless_calls_and_the_pool_bound(ea, oracle):
    """Two HOST THREADS inside mi355_msm() at once (ctypes drops the GIL): each leases its own context, ring and copy stream
    (csrc/msm_stateless.hpp:StatelessLease / ring_acquire; so far only raced under ThreadSanitizer against a fake copy engine,
    tests/tsan_pipeline.cpp) -- both results equal the oracle, several rounds, different sizes and curves in flight together.
    Afterwards the pool holds ONE idle context per (curve, device), not one per caller, mi355_msm_pool_stats reports
    it, and mi355_msm_trim() empties it."""
    import threading

    ea.trim()
    assert ea.pool_stats()["idle_contexts"] == 0
    jobs = []
    for t, (curve, cid, n) in enumerate((("bls12_377_g1", 0, 60001), ("bls12_377_g1", 0, 1 << 16), ("bls12_381_g1", 1, 40000), ("bls12_377_g1", 0, 3000))):
        bases = ea.generate_points(n, distinct=max(1, n // 5), seed=100 + t, curve=curve)
        sc = _scalars(n, 200 + t)
        jobs.append((curve, cid, n, bases, sc, oracle_msm_np(oracle, cid, bases, sc, n)))
    errors = []

    def worker(job, rounds):
        curve, cid, n, bases, sc, exp = job
        try:
            for _ in range(rounds):
                got = ea.msm(bases, sc, curve)
                if got != exp:
                    errors.append((curve, n, "result differs from the oracle"))
        except Exception as e:   # noqa: BLE001
            errors.append((curve, n, repr(e)))

    threads = [threading.Thread(target=worker, args=(job, 4)) for job in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not errors, errors
    st = ea.pool_stats()
    # three callers used bls12_377_g1 on this device at the same time, one bls12_381_g1: one parked context per key
    assert st["idle_contexts"] == 2, st
    assert st["idle_rings"] >= 1 and st["idle_pinned_bytes"] == st["idle_rings"] * 12 * (16 << 20)
    ea.trim()
    assert ea.pool_stats() == {"idle_contexts": 0, "idle_device_bytes": 0, "idle_rings": 0, "idle_pinned_bytes": 0}


def test_idle_context_gives_its_buffers_back_above_the_threshold(ea, oracle):
    """MI355_MSM_STATELESS_KEEP_MB bounds what a parked context may hold: 0 parks it without device buffers; the next call still works."""
    n = 50000
    bases = ea.generate_points(n, distinct=5000, seed=77)
    sc = _scalars(n, 78)
    exp = oracle_msm_np(oracle, 0, bases, sc, n)
    ea.trim()
    assert ea.msm(bases, sc) == exp
    held = ea.pool_stats()["idle_device_bytes"]
    assert held > n * 100                                   # bases, scalars, raw records and work buffers stay with the parked context
    with _Env(MI355_MSM_STATELESS_KEEP_MB=0):
        assert ea.msm(bases, sc) == exp
        st = ea.pool_stats()
        assert st["idle_contexts"] == 1 and st["idle_device_bytes"] < held // 4, (st, held)
        assert ea.msm(bases, sc) == exp
    ea.trim()


def test_out_of_memory_reclaims_the_idle_contexts_first(ea, oracle):
    """A run whose allocation fails with out-of-memory first frees the parked stateless contexts and tries the SAME chunk again;
    only when nothing is parked does it fall back to half the chunk (the inject_alloc_failures hook makes the failure
    deterministic; DevBuf::reserve applies the same rule to a real hipMalloc failure)."""
    n = 1 << 16
    bases = ea.generate_points(n, distinct=4096, seed=5)
    sc = _scalars(n, 6)
    exp = oracle_msm_np(oracle, 0, bases, sc, n)
    ea.trim()
    assert ea.msm(bases, sc) == exp
    assert ea.pool_stats()["idle_contexts"] == 1
    ctx = ea.multi_scalar_mult_init(bases, "bls12_377_g1")
    ctx.set_option("inject_alloc_failures", 1)
    assert ctx.run(sc)[0] == exp
    assert ea.pool_stats()["idle_contexts"] == 0, "the parked context should have been reclaimed by the failing allocation"
    assert ctx.query("oom_backoffs") == 0 and ctx.query("chunk_cap") == 0          # ... and the chunk was NOT halved
    ctx.set_option("inject_alloc_failures", 1)                                      # nothing parked any more: now the chunk halves
    assert ctx.run(sc)[0] == exp
    assert ctx.query("oom_backoffs") == 1
    ctx.close()


def test_concurrent_streaming_accumulators(ea, oracle):
    """Two ChunkedPippenger objects flushing from two host threads at once (every flush is a stateless pipeline run)."""
    import threading

    jobs = []
    for t in range(2):
        n = 7000 + 1500 * t
        bases = ea.generate_points(n, distinct=300, seed=40 + t)
        sc = _scalars(n, 50 + t)
        jobs.append((n, bases, sc, oracle_msm_np(oracle, 0, bases, sc, n)))
    errors = []

    def worker(job):
        n, bases, sc, exp = job
        try:
            cp = ea.ChunkedPippenger(1000)
            for lo in range(0, n, 777):
                cp.add(bases[lo:lo + 777], sc[lo:lo + 777])
            got = cp.finalize()
            flushes = cp.query("flushes")
            cp.close()
            if got != exp or flushes < n // 1000:
                errors.append((n, "mismatch", flushes))
        except Exception as e:   # noqa: BLE001
            errors.append((n, repr(e)))

    threads = [threading.Thread(target=worker, args=(j,)) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not errors, errors
    ea.trim()
