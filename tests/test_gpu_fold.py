"""GPU: the context option `assume_subgroup` -- the winners' top-bit trick (CMB ProcessSignedDigits.cu:10-20,123-128: if the top
bit of k is set use k' = r - k and -P), here for every k in (r/2, r).  Valid when r P = O for every base, which is what the ZPrize
generator produces; the option is off by default because arkworks' msm is exact for ANY curve point (tested last).
Bit-exact against the CPU oracle: the results are normalised projective images."""
import numpy as np
import pytest

import pymodel as m
from conftest import oracle_msm, oracle_msm_np

pytestmark = pytest.mark.gpu

CURVES = [(0, m.BLS12_377_G1), (1, m.BLS12_381_G1)]


def _scalars_with_edges(curve, n, seed):
    """uniform below r, with the values around the fold threshold, around r, and full 256-bit ones sprinkled in"""
    rng = np.random.default_rng(seed)
    r = curve.r
    ks = [int.from_bytes(rng.bytes(32), "little") % r for _ in range(n)]
    edges = [0, 1, 2, r - 1, r - 2, (r - 1) // 2, (r + 1) // 2, (r + 1) // 2 + 1, (r - 1) // 2 - 1, r, r + 1, 2 * r - 1, (1 << 256) - 1, (1 << 255) + 12345,
             1 << 252, (1 << 252) - 1, 1 << 253, (1 << 253) - 1, 1 << 254, (1 << 254) - 1, 3 << 250]
    for j, e in enumerate(edges):
        ks[(j * 37) % n] = e
    return ks


@pytest.mark.parametrize("cid,curve", CURVES)
def test_fold_equals_oracle_on_subgroup_points(ea, oracle, cid, curve):
    n = 6000
    bases = ea.generate_points(n, distinct=97, seed=21, curve=curve.name)   # h_j * G: in the order-r subgroup
    ks = _scalars_with_edges(curve, n, 5 + cid)
    # for points of order r the exact integer k and k mod r give the same sum: the oracle (arkworks' algorithm) takes k mod r
    scalars = np.frombuffer(m.encode_scalars([k % curve.r for k in ks]), dtype=np.uint8).reshape(n, 32)
    exp = oracle_msm_np(oracle, cid, bases, np.ascontiguousarray(scalars), n)
    raw = np.frombuffer(m.encode_scalars(ks), dtype=np.uint8).reshape(n, 32)
    ctx = ea.multi_scalar_mult_init(bases, curve.name)
    assert ea.multi_scalar_mult(ctx, bases, np.ascontiguousarray(raw))[0] == exp   # off: exact integers, same sum on these points
    ctx.set_option("assume_subgroup", 1)
    # windows that tile the folded bit length exactly (252 = 12 x 21 = 14 x 18 = 18 x 14 = 28 x 9; 254 = 2 x 127), and some that do not
    for c in (0, 9, 14, 18, 21, 2, 7, 11, 16, 20):
        ctx.set_option("window_bits", c)
        assert ea.multi_scalar_mult(ctx, bases, np.ascontiguousarray(raw))[0] == exp, (curve.name, c)
    ctx.set_option("window_bits", 0)
    # chunked runs and the Fr-Montgomery entry go through the same digit code
    ctx.set_option("max_chunk", 1500)
    assert ea.multi_scalar_mult(ctx, bases, np.ascontiguousarray(raw))[0] == exp
    ctx.set_option("max_chunk", 0)
    mont = np.frombuffer(m.encode_scalars([(k % curve.r) * (1 << 256) % curve.r for k in ks]), dtype=np.uint8).reshape(n, 32)
    ctx.set_option("scalars_montgomery", 1)
    assert ea.multi_scalar_mult(ctx, bases, np.ascontiguousarray(mont))[0] == exp
    ctx.close()


def test_fold_g2_and_tables(ea, oracle):
    c = m.BLS12_377_G2
    n = 1500
    bases = ea.generate_points(n, distinct=50, seed=4, curve=c.name)
    ks = [k % c.r for k in _scalars_with_edges(c, n, 9)]
    sc = np.frombuffer(m.encode_scalars(ks), dtype=np.uint8).reshape(n, 32)
    out = np.zeros(288, dtype=np.uint8)
    assert oracle.oracle_msm(2, bases.ctypes.data, 200, sc.ctypes.data, n, out.ctypes.data, 0) == 0
    ctx = ea.multi_scalar_mult_init(bases, c.name)
    ctx.set_option("assume_subgroup", 1)
    for w in (0, 14, 18):
        ctx.set_option("window_bits", w)
        assert ea.multi_scalar_mult(ctx, bases, np.ascontiguousarray(sc))[0] == out.tobytes(), w
    ctx.close()
    # precomputed tables (all windows share one bucket set) with folded scalars
    g1 = m.BLS12_377_G1
    bases = ea.generate_points(4096, distinct=64, seed=2, curve=g1.name)
    ks = [k % g1.r for k in _scalars_with_edges(g1, 4096, 3)]
    sc = np.frombuffer(m.encode_scalars(ks), dtype=np.uint8).reshape(4096, 32)
    ctx = ea.MultiScalarMultContext(g1.name)
    ctx.set_option("precompute", 1)
    ctx.set_option("assume_subgroup", 1)
    ctx.set_bases(bases)
    assert ctx.run(np.ascontiguousarray(sc))[0] == oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(sc), 4096)
    ctx.close()


def test_default_stays_exact_outside_the_subgroup(ea, oracle):
    """Why the option is not the default: for a curve point outside the order-r subgroup k P != (r - k)(-P).  The default path
    equals the oracle on such points (arkworks adds whatever points it is given); with the option it must not be used."""
    curve = m.BLS12_377_G1
    T = (curve.p - 1, 0)   # the 2-torsion point of y^2 = x^3 + 1 (the FPGA harness's edge fixture, P1B hardcaml msm_unit_tests.rs:30-49)
    assert curve.on_curve(T) and curve.add(T, T) is None
    G = curve.generator()
    pts = [curve.add(curve.mul(3 + i, G), T) for i in range(7)]   # order 2r: on the curve, outside the subgroup
    ks = [curve.r - ((i + 1) << 225) for i in range(7)]   # above r/2 with a top limb below top(r): folded; an odd count: the seven 2-torsion parts do not cancel
    bases, scalars = curve.encode_affine_array(pts), m.encode_scalars(ks)
    exp = oracle_msm(oracle, 0, bases, scalars, 7)
    assert ea.msm(bases, scalars, curve.name) == exp
    ctx = ea.multi_scalar_mult_init(bases, curve.name)
    assert ea.multi_scalar_mult(ctx, bases, scalars)[0] == exp
    ctx.set_option("assume_subgroup", 1)
    assert ea.multi_scalar_mult(ctx, bases, scalars)[0] != exp   # the documented difference
    ctx.close()


def test_environment_switch_for_harness_contexts(ea, oracle, monkeypatch):
    """mi355_msm_create_env (what every harness shim calls): MI355_MSM_ASSUME_SUBGROUP sets the option; unset leaves it off."""
    curve = m.BLS12_377_G1
    n = 700
    bases = ea.generate_points(n, distinct=33, seed=1, curve=curve.name)
    ks = _scalars_with_edges(curve, n, 2)
    sc = np.frombuffer(m.encode_scalars([k % curve.r for k in ks]), dtype=np.uint8).reshape(n, 32)
    exp = oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(sc), n)
    for env, want in ((None, 0), ("1", 1), ("0", 0)):
        if env is None:
            monkeypatch.delenv("MI355_MSM_ASSUME_SUBGROUP", raising=False)
        else:
            monkeypatch.setenv("MI355_MSM_ASSUME_SUBGROUP", env)
        ctx = ea.MultiScalarMultContext.from_env(curve.name)
        assert ctx.query("assume_subgroup") == want and ctx.query("carry") == 1
        ctx.set_bases(bases)
        assert ctx.run(np.ascontiguousarray(sc))[0] == exp
        ctx.close()


def test_environment_switch_for_precompute(ea, oracle, monkeypatch):
    """MI355_MSM_PRECOMPUTE = auto | 1 | 0 and MI355_MSM_TABLE_LEVELS through mi355_msm_create_env (the harness shims' constructor):
    the reference's init builds its tables untimed (CMB MSM.cu:380-383), a harness that owns the GPU opts in through the environment."""
    curve = m.BLS12_377_G1
    n = 5000
    bases = ea.generate_points(n, distinct=77, seed=3, curve=curve.name)
    sc = np.frombuffer(m.encode_scalars([k % curve.r for k in _scalars_with_edges(curve, n, 4)]), dtype=np.uint8).reshape(n, 32)
    sc = np.ascontiguousarray(sc)
    exp = oracle_msm_np(oracle, 0, bases, sc, n)
    for pre, levels, want_pre, want_tables in ((None, None, 0, False), ("auto", None, 2, False), ("1", "3", 1, True), ("0", None, 0, False)):
        for k, v in (("MI355_MSM_PRECOMPUTE", pre), ("MI355_MSM_TABLE_LEVELS", levels)):
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, v)
        ctx = ea.MultiScalarMultContext.from_env(curve.name)
        assert ctx.query("precompute") == want_pre
        ctx.set_bases(bases)
        assert (ctx.query("table_levels") > 1) == want_tables       # (auto builds none for an input this small)
        if want_tables:
            assert ctx.query("table_levels") <= 3
        assert ctx.run(sc)[0] == exp
        ctx.close()
    monkeypatch.setenv("MI355_MSM_PRECOMPUTE", "auto")
    monkeypatch.setenv("MI355_MSM_TABLE_LEVELS", "999")
    with pytest.raises(ea.MsmError):
        ea.MultiScalarMultContext.from_env(curve.name)


@pytest.mark.parametrize("cid,curve", CURVES)
def test_every_scalar_folded_and_sharded_contexts(ea, oracle, cid, curve):
    """All scalars in (r/2, r) -- every one of them runs as (r - k)(-P) -- and none (all below r/2); tiny inputs; a sharded context
    (the option is forwarded to every shard)."""
    rng = np.random.default_rng(17 + cid)
    r = curve.r
    for n in (1, 2, 64, 3001):
        bases = ea.generate_points(n, distinct=min(n, 41), seed=n, curve=curve.name)
        hi = [r // 2 + 1 + int.from_bytes(rng.bytes(32), "little") % (r // 2 - 1) for _ in range(n)]
        lo = [int.from_bytes(rng.bytes(32), "little") % (r // 2) for _ in range(n)]
        for ks in (hi, lo):
            sc = np.frombuffer(m.encode_scalars(ks), dtype=np.uint8).reshape(n, 32)
            exp = oracle_msm_np(oracle, cid, bases, np.ascontiguousarray(sc), n)
            for devices in (None, [0, 0, 0]):
                ctx = ea.multi_scalar_mult_init(bases, curve.name, devices=devices) if devices else ea.multi_scalar_mult_init(bases, curve.name)
                ctx.set_option("assume_subgroup", 1)
                assert ctx.query("assume_subgroup") == 1
                assert ctx.run(np.ascontiguousarray(sc))[0] == exp, (n, devices)
                ctx.close()
