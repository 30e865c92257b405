"""Wall time of the stateless mi355_msm() call at 2^npow pairs for a few staging-thread counts and slice sizes
(profiles/r03_stateless_probe.txt).  Operands live in ordinary (pageable) numpy arrays."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import entries_amd as ea  # noqa: E402

npow = int(sys.argv[1]) if len(sys.argv) > 1 else 26
curve = sys.argv[2] if len(sys.argv) > 2 else "bls12_377_g1"
n = 1 << npow
distinct = min(n, 1 << 15)
tile = ea.generate_points(distinct, distinct=distinct, seed=5, curve=curve)
bases = np.ascontiguousarray(np.tile(tile, (n // distinct, 1)))
rng = np.random.default_rng(1)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
sc[:, 31] &= 0x0F
print(f"{curve} 2^{npow}: {bases.nbytes / 1e9:.2f} GB bases + {sc.nbytes / 1e9:.2f} GB scalars; PCIe floor at 57 GB/s = {(bases.nbytes + sc.nbytes) / 57e6:.1f} ms")
ref = None
for label, env in (("default (first call: ring allocation)", {}), ("default", {}), ("default", {}),
                   ("threads 2", {"MI355_MSM_STAGE_THREADS": 2}), ("threads 4", {"MI355_MSM_STAGE_THREADS": 4}),
                   ("threads 8", {"MI355_MSM_STAGE_THREADS": 8}), ("threads 12", {"MI355_MSM_STAGE_THREADS": 12}),
                   ("slice 2^22", {"MI355_MSM_STATELESS_SLICE_LOG": 22}), ("slice 2^24", {"MI355_MSM_STATELESS_SLICE_LOG": 24}),
                   ("slice 2^21", {"MI355_MSM_STATELESS_SLICE_LOG": 21}), ("one slice (serial)", {"MI355_MSM_STATELESS_SLICE_LOG": 26})):
    for k, v in env.items():
        os.environ[k] = str(v)
    t0 = time.perf_counter()
    r = ea.msm(bases, sc, curve)
    dt = (time.perf_counter() - t0) * 1e3
    for k in env:
        os.environ.pop(k)
    st = ea.last_stateless()
    ref = ref or r
    print(f"{label:40s} {dt:8.1f} ms   setup {st['setup_ms']:6.1f}  waited-for-upload {st['wait_upload_ms']:6.1f}  compute {st['compute_ms']:6.1f} "
          f"(last slice {st['tail_ms']:5.1f})  slices {int(st['slices'])} threads {int(st['threads'])}  same={r == ref}")
# fresh pages
b2, s2 = bases.copy(), sc.copy()
t0 = time.perf_counter()
r = ea.msm(b2, s2, curve)
print(f"{'fresh operands (never seen by HIP)':40s} {(time.perf_counter() - t0) * 1e3:8.1f} ms   same={r == ref}")
# the context path on the same operands for comparison
ctx = ea.multi_scalar_mult_init(bases, curve)
t0 = time.perf_counter()
rc = ctx.run(sc)[0]
print(f"{'context path, host scalars':40s} {(time.perf_counter() - t0) * 1e3:8.1f} ms   same={rc == ref}")
ctx.close()
