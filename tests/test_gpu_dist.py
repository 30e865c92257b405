"""GPU: the N > 1 path end to end with real device MSMs -- two ranks (sharing the one GPU of the test box, gloo for the
144-byte exchange; on a multi-GPU node the same code runs one rank per GPU over RCCL): disjoint base/scalar slices,
all-gather of the partials, fold on every rank, equal to the CPU oracle on the whole input."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, oracle_msm_np

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, curve, bases, scalars, n, q):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import entries_amd as ea

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lo, hi = ea.shard_bounds(n, world, rank)
    ctx = ea.multi_scalar_mult_init(torch.from_numpy(bases[lo:hi]).cuda(), curve)
    d_scalars = torch.from_numpy(scalars[lo:hi]).cuda()
    res = ea.sharded_msm(lambda: ctx.run(d_scalars)[0], curve)
    q.put((rank, res))
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("curve,cid,n", [("bls12_377_g1", 0, 50001), ("bls12_381_g1", 1, 1 << 15)])
def test_two_ranks_shard_gather_fold(ea, oracle, curve, cid, n):
    bases = ea.generate_points(n, distinct=777, seed=3, curve=curve)
    rng = np.random.default_rng(n)
    scalars = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    scalars[:, 31] &= 0x0F
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() + n) % 1500
    procs = [ctx.Process(target=_worker, args=(r, 2, port, curve, bases, scalars, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0] == results[1] == oracle_msm_np(oracle, cid, bases, scalars, n)
