"""On the GPU box: the stateless call against the shape of its slices (profiles/r03_stateless_shape.txt):
slice size, a ramp-down at the end (slice/2, /4, /8), carried buckets across the slices."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import entries_amd as ea
import bench

n = 1 << 26
curve = "bls12_377_g1"
dev = torch.device("cuda", 0)
tile = torch.from_numpy(ea.generate_points(1 << 15, distinct=1 << 15, seed=1, curve=curve)).to(dev)
bases_host = tile.repeat(n >> 15, 1).contiguous().cpu().numpy()
sc_host = bench.uniform_scalars(n, bench.R377_TOP, dev, 7).cpu().numpy()
ref = None
shapes = ((23, 0, 0, 0), (23, 1, 0, 0), (24, 0, 0, 0), (24, 1, 0, 0), (25, 1, 0, 0), (23, 1, 1, 0), (24, 1, 1, 0), (23, 0, 0, 0))
if len(sys.argv) > 1 and sys.argv[1] == "carry_c":   # carried slices at a forced window size (MI355_MSM_STATELESS_CARRY_C)
    shapes = ((23, 0, 0, 0), (23, 0, 1, 17), (23, 0, 1, 18), (23, 0, 1, 19), (23, 1, 1, 18), (23, 1, 1, 19), (23, 0, 1, 20), (23, 0, 0, 0))
for slog, down, carry, cc in shapes:
    os.environ["MI355_MSM_STATELESS_CARRY_C"] = str(cc)
    os.environ["MI355_MSM_STATELESS_SLICE_LOG"] = str(slog)
    os.environ["MI355_MSM_STATELESS_RAMP_DOWN"] = str(down)
    os.environ["MI355_MSM_STATELESS_CARRY"] = str(carry)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); r = ea.msm(bases_host, sc_host, curve); best = min(best, time.perf_counter() - t0)
    ref = ref or r
    st = ea.last_stateless()
    print("slice 2^%d ramp-down %d carry %d (c %s): %.1f ms  same=%s  waited-for-upload %.1f  issue/await %.1f (after the last upload %.1f)  last DMA done @%.1f  slices %d" % (
        slog, down, carry, cc or "auto", best * 1e3, r == ref, st["wait_upload_ms"], st["compute_ms"], st["tail_ms"], st.get("dma_done_ms", 0.0), st["slices"]), flush=True)
