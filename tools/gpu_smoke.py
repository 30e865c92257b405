"""Quick GPU check used during bring-up: small MSMs against oracle/pymodel.py, then timing sweeps."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import torch
import entries_amd as ea
import pymodel as m

def check(curve, n, rng, distinct=None, scalars=None, pts=None, label=""):
    pts = pts if pts is not None else m.random_points(curve, n, rng, distinct)
    sc = scalars if scalars is not None else m.random_scalars(curve, n, rng)
    ctx = ea.multi_scalar_mult_init(curve.encode_affine_array(pts), curve.name)
    got = ea.multi_scalar_mult(ctx, curve.encode_affine_array(pts), m.encode_scalars(sc))[0]
    exp = curve.encode_projective_normalized(curve.msm_pippenger(pts, sc) if n > 64 else curve.msm_naive(pts, sc))
    ok = got == exp
    print(f"{curve.name} n={n} {label} -> {'OK' if ok else 'MISMATCH'}  {ctx.last_timings()}", flush=True)
    ctx.close()
    return ok

def main():
    rng = random.Random(11)
    allok = True
    for curve in (m.BLS12_377_G1, m.BLS12_381_G1):
        for n in (1, 33, 1024):
            allok &= check(curve, n, rng, distinct=max(1, n // 4))
        # all-equal scalars: one hot bucket per window
        pts = m.random_points(curve, 200, rng, 7)
        allok &= check(curve, 200, rng, pts=pts, scalars=[12345678901234567890] * 200, label="hot-bucket")
        allok &= check(curve, 200, rng, pts=pts, scalars=[1] * 200, label="unit-scalars")
        allok &= check(curve, 200, rng, pts=pts, scalars=[(1 << 256) - 1] * 200, label="max-scalars")
        pts2 = list(pts); pts2[3] = None; pts2[100] = None
        allok &= check(curve, 200, rng, pts=pts2, label="with-infinity")
    c = m.BLS12_377_G1
    allok &= check(c, 4, rng, pts=[m.EDGE_P, m.EDGE_P_NEG, m.EDGE_T, m.EDGE_T], scalars=[5] * 4, label="edge1")
    allok &= check(c, 2, rng, pts=[c.generator(), m.EDGE_P], scalars=[1, 2], label="edge5")
    print("ALL OK" if allok else "SOME MISMATCH", flush=True)
    # timing sweep on replicated random data (no check)
    import torch
    pts = m.random_points(c, 1024, rng)
    img = np.frombuffer(c.encode_affine_array(pts), dtype=np.uint8).reshape(1024, 104)
    for lg in (16, 20, 22, 24):
        n = 1 << lg
        bases = torch.from_numpy(np.tile(img, (n // 1024, 1))).cuda()
        sc = torch.from_numpy(np.random.default_rng(lg).integers(0, 256, size=(n, 32), dtype=np.uint8))
        sc[:, 31] &= 0x0F
        sc = sc.cuda()
        ctx = ea.MultiScalarMultContext("bls12_377_g1")
        ctx.set_bases(bases)
        for rep in range(2):
            t0 = time.time(); r = ctx.run(sc); dt = time.time() - t0
            print(f"n=2^{lg} wall={dt*1e3:.1f} ms  {ctx.last_timings()}", flush=True)
        ctx.close()

if __name__ == "__main__":
    main()
