"""GPU: the sharded context behind the C ABI (mi355_msm_create_sharded).  One box has one MI355X, so the shards here are
LOGICAL -- the same device listed several times -- which exercises everything but the xGMI hop: per-shard host threads,
contexts and streams, slice bounds, strided batches, prefix runs, and the fold.  The RCCL all-gather is exercised with a
one-rank communicator (devices = [0], combine = 2).  The full BASELINE config 4 -- 2^28 pairs as 8 shards of 2^25 -- runs
against the single-context chunked result and an oracle-checked prefix."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT, oracle_msm_np

pytestmark = pytest.mark.gpu

R377_TOP = 0x12ab655e9a2ca556


def _scalars(n, seed):
    rng = np.random.default_rng(seed)
    limbs = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] %= np.uint64(R377_TOP)
    return limbs.view(np.uint8).reshape(n, 32)


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_logical_shards_match_oracle(ea, oracle, G):
    import torch

    n, batches = 5003, 2          # not a multiple of any G: ragged last shard
    bases = ea.generate_points(n, distinct=611, seed=21)
    sc = _scalars(batches * n, 5 + G)
    sc[7] = 0
    exp = [oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(sc[b * n:(b + 1) * n]), n) for b in range(batches)]
    ctx = ea.MultiScalarMultContext("bls12_377_g1", devices=[0] * G)
    assert ctx.query("shards") == G
    ctx.set_bases(bases)
    assert ctx.query("bases") == n
    assert ctx.run(sc) == exp                                             # host scalars, batch stride = n
    assert ctx.run(torch.from_numpy(sc).cuda()) == exp                    # device scalars, read in place by every shard
    # a prefix run only involves the shards that own part of the prefix
    k = 1300
    assert ctx.run(np.ascontiguousarray(sc[:k]), npoints=k)[0] == oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(sc[:k]), k)
    # bases handed over as ONE device buffer
    ctx.set_bases(torch.from_numpy(bases).cuda())
    assert ctx.run(sc) == exp
    tm = ctx.last_timings()
    assert tm["accumulate"] > 0 and tm["window_bits"] > 0
    ctx.close()


def test_more_shards_than_points_and_empty(ea, oracle):
    bases = ea.generate_points(3, distinct=3, seed=2)
    sc = _scalars(3, 1)
    ctx = ea.MultiScalarMultContext("bls12_377_g1", devices=[0] * 8)
    ctx.set_bases(bases)
    assert ctx.run(sc)[0] == oracle_msm_np(oracle, 0, bases, sc, 3)
    ctx.set_bases(np.zeros((0, 104), dtype=np.uint8))
    inf = ctx.run(np.zeros((0, 32), dtype=np.uint8))[0]
    assert inf[96:] == bytes(48)        # (1, 1, 0)
    ctx.close()


def test_other_curves_and_options_forwarded(ea, oracle):
    for curve, cid in (("bls12_381_g1", 1), ("bls12_377_g2", 2)):
        n = 700
        bases = ea.generate_points(n, distinct=90, seed=4, curve=curve)
        sc = _scalars(n, 9)
        sc[:, 31] &= 0x0F
        ctx = ea.MultiScalarMultContext(curve, devices=[0, 0, 0])
        ctx.set_option("window_bits", 9)
        ctx.set_bases(bases)
        out = ctypes.create_string_buffer(ea.projective_bytes(curve))
        assert oracle.oracle_msm(cid, bases.ctypes.data, ea.affine_stride(curve), sc.ctypes.data, n, out, 0) == 0
        assert ctx.run(sc)[0] == out.raw
        assert ctx.last_timings()["window_bits"] == 9
        ctx.close()


def test_rccl_all_gather_path_one_rank(ea, oracle):
    """devices = [0]: a one-rank RCCL communicator -- ncclCommInitAll, ncclAllGather inside a group, the D2H of the gathered
    buffer and the byte-compare with the host copy all execute; combine = 2 makes a missing librccl an error, not a fallback."""
    n = 2000
    bases = ea.generate_points(n, distinct=300, seed=8)
    sc = _scalars(2 * n, 3)
    ctx = ea.MultiScalarMultContext("bls12_377_g1", devices=[0])
    ctx.set_option("combine", 2)
    ctx.set_bases(bases)
    got = ctx.run(sc)
    assert got == [oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(sc[b * n:(b + 1) * n]), n) for b in range(2)]
    assert ctx.query("rccl_exchanges") == 1
    ctx.close()
    # logical shards cannot form a communicator: auto falls back to the host fold, "require" says why
    ctx = ea.MultiScalarMultContext("bls12_377_g1", devices=[0, 0])
    ctx.set_bases(bases)
    assert ctx.run(np.ascontiguousarray(sc[:n]))[0] == got[0] and ctx.query("rccl_exchanges") == 0
    ctx.set_option("combine", 2)
    with pytest.raises(ea.MsmError) as ei:
        ctx.run(np.ascontiguousarray(sc[:n]))
    assert "distinct devices" in ei.value.message
    ctx.close()


def test_harness_shim_honours_devices_env(ea, oracle):
    """The ZPrize harness FFI names with MI355_MSM_DEVICES set: the unchanged harness runs sharded."""
    import subprocess
    import sys

    code = r'''
import ctypes, os, sys
sys.path.insert(0, %r)
import numpy as np
import entries_amd as ea
class RustError(ctypes.Structure):
    _fields_ = [("code", ctypes.c_int), ("message", ctypes.c_char_p)]
lib = ctypes.CDLL(os.path.join(ea.PACKAGE_DIR, "libmi355msm_zprize_377.so"))
lib.mult_pippenger_init.restype = RustError
lib.mult_pippenger_inf.restype = RustError
n = 4096
bases = ea.generate_points(n, distinct=500, seed=77)
rng = np.random.default_rng(1)
sc = rng.integers(0, 256, size=(2 * n, 32), dtype=np.uint8); sc[:, 31] &= 0x0f
ctx = ctypes.c_void_p()
e = lib.mult_pippenger_init(ctypes.byref(ctx), bases.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), ctypes.c_size_t(104))
assert e.code == 0, e.message
out = ctypes.create_string_buffer(288)
e = lib.mult_pippenger_inf(ctypes.byref(ctx), out, bases.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), ctypes.c_size_t(2),
                           sc.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(104))
assert e.code == 0, e.message
v = ctypes.c_uint64()
ea.load_library().mi355_msm_query(ctx, b"shards", ctypes.byref(v))
sys.stdout.write("SHARDS %%d\n" %% v.value)
sys.stdout.write(out.raw.hex() + "\n")
np.save(sys.argv[1], sc)
''' % ROOT
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, MI355_MSM_DEVICES="0,0,0,0")
        r = subprocess.run([sys.executable, "-c", code, os.path.join(d, "sc.npy")], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = r.stdout.strip().splitlines()
        assert lines[-2] == "SHARDS 4"
        sc = np.load(os.path.join(d, "sc.npy"))
    n = 4096
    bases = ea.generate_points(n, distinct=500, seed=77)
    exp = b"".join(oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(sc[b * n:(b + 1) * n]), n) for b in range(2))
    assert bytes.fromhex(lines[-1]) == exp


def test_full_2_28_as_8_shards_of_2_25(ea, oracle):
    """BASELINE.json configs[3] on the one GPU there is: 2^28 pairs, 8 logical shards of 2^25 (devices = {0,...,0}), against
    (i) one context running the same 2^28 pairs as four chunks of 2^26, (ii) the oracle on a 2^16 prefix, and
    (iii) linearity: the sum of the two halves run as separate prefix/suffix problems is not needed -- (i) already compares
    two different decompositions of the same sum bit for bit."""
    import torch

    n, distinct = 1 << 28, 1 << 15
    tile_np = ea.generate_points(distinct, distinct=distinct, seed=0x5A5052495A45)
    tile = torch.from_numpy(tile_np).cuda()
    bases = tile.repeat(n // distinct, 1).contiguous()           # 27.9 GB on the device
    g = torch.Generator(device="cuda")
    g.manual_seed(2028)
    limbs = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    limbs[:, 3] &= (1 << 59) - 1                                  # < 2^251 < r: canonical scalars
    scalars = limbs.view(torch.uint8).reshape(n, 32)
    del limbs
    # (i) one context, chunked
    one = ea.MultiScalarMultContext("bls12_377_g1")
    one.set_bases(bases)
    whole = one.run(scalars)[0]
    assert one.last_timings()["launches"] >= 4
    k = 1 << 16
    prefix_one = one.run(scalars[:k].contiguous(), npoints=k)[0]
    one.close()
    # (ii) eight logical shards of 2^25
    sh = ea.MultiScalarMultContext("bls12_377_g1", devices=[0] * 8)
    sh.set_bases(bases)
    del bases
    torch.cuda.empty_cache()
    assert sh.query("shards") == 8 and sh.query("bases") == n
    lo, hi = ea.shard_bounds(n, 8, 5)
    assert hi - lo == 1 << 25
    got = sh.run(scalars)[0]
    assert got == whole
    assert sh.query("twisted_edwards") == 1
    prefix_sh = sh.run(scalars[:k].contiguous(), npoints=k)[0]
    sh.close()
    sc_np = scalars[:k].cpu().numpy()
    bases_np = np.ascontiguousarray(np.tile(tile_np, (k // distinct, 1)))
    exp = oracle_msm_np(oracle, 0, bases_np, sc_np, k)
    assert prefix_one == exp and prefix_sh == exp


@pytest.mark.parametrize("curve,cid", [("bls12_377_g1", 0), ("bls12_377_g2", 2)])
def test_peer_staging_branches_on_logical_shards(ea, oracle, curve, cid):
    """The two code paths of a sharded context that pull a shard's slice out of ANOTHER device's memory -- the base staging of
    sharded_set_bases and the per-batch scalar pull of sharded_run (csrc/msm_sharded.hpp) -- cannot be reached with logical shards
    (the source always lies on the shard's own device), so they would first execute on the first multi-GPU box, under the driver's
    clock.  The test hook "force_peer_staging" takes them on one GPU (the copies are then device-local; everything else is the peer
    code): BASELINE configs[3]'s shape -- 8 shards, device-resident bases and scalars, a ragged size, two batches, a prefix run --
    against the oracle, with the branch count asserted through mi355_msm_query "peer_stagings"."""
    import ctypes

    import torch

    stride = ea.affine_stride(curve)
    n = 8 * 1500 + 37
    bases = ea.generate_points(n, distinct=400, seed=31, curve=curve)
    sc = np.ascontiguousarray(np.random.default_rng(5).integers(0, 256, size=(2 * n, 32), dtype=np.uint8))
    sc[:, 31] &= 0x0F
    exp = []
    for b in range(2):
        out = ctypes.create_string_buffer(ea.projective_bytes(curve))
        part = np.ascontiguousarray(sc[b * n:(b + 1) * n])
        assert oracle.oracle_msm(cid, bases.ctypes.data, stride, part.ctypes.data, n, out, 0) == 0
        exp.append(out.raw)
    d_bases, d_sc = torch.from_numpy(bases).cuda(), torch.from_numpy(sc).cuda()
    ctx = ea.MultiScalarMultContext(curve, devices=[0] * 8)
    assert ctx.query("peer_stagings") == 0
    ctx.set_bases(d_bases)
    assert ctx.run(d_sc) == exp and ctx.query("peer_stagings") == 0        # the ordinary (own-device) branches
    ctx.set_option("force_peer_staging", 1)
    ctx.set_bases(d_bases)
    assert ctx.query("peer_stagings") == 8                                   # one staged slice of bases per shard
    assert ctx.run(d_sc) == exp
    assert ctx.query("peer_stagings") == 16                                  # ... and one scalar pull per shard and run
    k = 5 * 1500 + 3                                                         # a prefix: the last shards get nothing
    out = ctypes.create_string_buffer(ea.projective_bytes(curve))
    part = np.ascontiguousarray(sc[:k])
    assert oracle.oracle_msm(cid, bases.ctypes.data, stride, part.ctypes.data, k, out, 0) == 0
    assert ctx.run(d_sc[:k].contiguous(), npoints=k)[0] == out.raw
    assert 16 < ctx.query("peer_stagings") <= 24
    ctx.close()
