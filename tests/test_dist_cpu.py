"""CPU, world_size 2, gloo: the N > 1 path -- disjoint slices, all-gather of 144-byte partials, fold -- with the per-rank
MSM stood in by the CPU oracle (there is no GPU here; on the GPU box the same dist code runs over RCCL)."""
import os
import random
import sys

import pytest
import torch.multiprocessing as mp

import pymodel as m
from conftest import ROOT


def _worker(rank, world, port, curve_name, bases, scalars, n, q):
    import ctypes

    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import entries_amd as ea

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    cid = ea.CURVE_IDS[curve_name]
    lo, hi = ea.shard_bounds(n, world, rank)

    def local():
        out = ctypes.create_string_buffer(144)
        b = bases[lo * 104:hi * 104]
        s = scalars[lo * 32:hi * 32]
        lib.oracle_msm(cid, ctypes.create_string_buffer(b, len(b) or 1), ctypes.c_size_t(104),
                       ctypes.create_string_buffer(s, len(s) or 1), ctypes.c_size_t(hi - lo), out, 1)
        return out.raw

    res = ea.sharded_msm(local, curve_name)
    q.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.parametrize("curve,n", [(m.BLS12_377_G1, 101), (m.BLS12_381_G1, 64), (m.BLS12_377_G1, 1)])
def test_sharded_msm_world2(built, oracle, curve, n):
    from conftest import oracle_msm

    rng = random.Random(n)
    pts = m.random_points(curve, n, rng, max(1, n // 4))
    sc = m.random_scalars(curve, n, rng)
    bases, scalars = curve.encode_affine_array(pts), m.encode_scalars(sc)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, curve.name, bases, scalars, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = oracle_msm(oracle, curve.curve_id, bases, scalars, n)
    assert results[0] == results[1] == expected
