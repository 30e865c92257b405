"""GPU: the anchored window (csrc/msm_engine.hip "the anchored window", csrc/partition.hpp next_digit).

Signed digits carry, so the window above a canonical scalar's last FULL window is non-zero for 14-57 % of the scalars even where the window
size leaves it (almost) no bits of its own -- BLS12-377's Fr: 253 = 11 x 23, 252 = 12 x 21.  The ZPrize winners remove those additions by
halving the scalar (CMB ProcessSignedDigits.cu:10-20,123-128), which needs r P = O.  Here the carry chain ENDS at the last full window: its
value v in [0, 2^c] is taken as 2^(c-1) + s, |s| <= 2^(c-1), and the constant part -- 2^(c a + c - 1) times the plain sum of the bases --
is added on the host; that sum is computed by the pipeline itself the first time a context runs a given number of pairs.  No assumption
on the inputs: the results below are the oracle's bytes for zero / one / maximal / non-canonical scalars, bases at infinity, points
outside the prime-order subgroup, chunked (carried) batches, host and device scalars, precomputed tables, every curve.
Option "anchor": 0 off, 1 (default) batches of 2^20 pairs and more at the window sizes where it saves >= 1 % of the additions,
2 = always (the test setting: any size, any window size)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R = {0: 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001,
     1: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001}
NAMES = {0: "bls12_377_g1", 1: "bls12_381_g1", 2: "bls12_377_g2", 3: "bls12_381_g2"}
BITS = {0: 253, 1: 255, 2: 253, 3: 255}
STRIDE = {0: 104, 1: 104, 2: 200, 3: 200}


def oracle_msm_np(oracle, cid, bases, sc, n):
    out = np.zeros(288 if cid >= 2 else 144, dtype=np.uint8)
    bases, sc = np.ascontiguousarray(bases), np.ascontiguousarray(sc)
    assert oracle.oracle_msm(cid, bases.ctypes.data, STRIDE[cid], sc.ctypes.data, n, out.ctypes.data, 0) == 0
    return out.tobytes()


def _scalars(cid, n, seed, special=True):
    rng = np.random.default_rng(seed)
    r = R[cid & 1]
    limbs = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] %= np.uint64(r >> 192)
    sc = limbs.view(np.uint8).reshape(n, 32).copy()
    if special:
        # (the oracle keeps arkworks' truncation -- bits from ceil(bits / c) c on are dropped, ARK variable_base/mod.rs:118-124 -- so the
        #  values it is compared on stay below 2^bits; larger ones: test_any_256_bit_scalar)
        top = 1 << BITS[cid]
        vals = [0, 0, 1, 2, r - 1, r - 2, r, r + 1, top - 1, top >> 1, (top >> 1) - 1, top >> 2, r >> 1, (r >> 1) + 1]
        for c in (7, 11, 13, 16, 18, 21, 23):      # the anchored window at exactly 0, 2^(c-1) (s = 0: no entry), 2^c - 1, and a carry INTO it
            a = BITS[cid] // c - 1
            for v in (0, 1 << (c - 1), (1 << c) - 1):
                vals.append(v << (c * a))
            vals.append((((1 << c) - 1) << (c * a)) | (1 << (c * a - 1)))
            vals.append((1 << (c * a)) - 1)
        for i, v in enumerate(vals):
            sc[3 + 5 * i] = np.frombuffer(int(v % top).to_bytes(32, "little"), dtype=np.uint8)
    return sc


def test_any_256_bit_scalar(ea, oracle):
    """The library takes a scalar as the 256-bit integer it is (the windows above the anchored one run a fresh signed chain): on bases of
    the prime-order subgroup k P = (k mod r) P, which the oracle computes."""
    n = 6000
    rng = np.random.default_rng(99)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    for i, v in enumerate(((1 << 256) - 1, 1 << 255, 1 << 254, 1 << 253, (1 << 256) - (1 << 252), R[0] << 3, 0)):
        sc[7 * i] = np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)
    for cid in (0, 1):
        red = np.zeros_like(sc)
        for i in range(n):
            red[i] = np.frombuffer((int.from_bytes(sc[i].tobytes(), "little") % R[cid]).to_bytes(32, "little"), dtype=np.uint8)
        bases = ea.generate_points(n, distinct=128, seed=60 + cid, curve=NAMES[cid])
        exp = oracle_msm_np(oracle, cid, bases, red, n)
        ctx = ea.multi_scalar_mult_init(bases, NAMES[cid])
        for anchor in (0, 2):
            ctx.set_option("anchor", anchor)
            for c in (0, 9, 12, 21, 23):
                ctx.set_option("window_bits", c)
                assert ctx.run(sc)[0] == exp, (cid, anchor, c)
        ctx.close()


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_anchored_window_returns_the_oracles_bytes(ea, oracle, cid):
    import torch

    n = 9000 if cid < 2 else 3000
    bases = ea.generate_points(n, distinct=257, seed=31 + cid, curve=NAMES[cid])
    bases[11, -8] = 1            # bases at infinity: neither in the entries nor in the sum of the bases
    bases[n - 1, -8] = 1
    sc = _scalars(cid, 2 * n, 5 + cid)
    exp = [oracle_msm_np(oracle, cid, bases, np.ascontiguousarray(sc[b * n:(b + 1) * n]), n) for b in range(2)]
    ctx = ea.multi_scalar_mult_init(bases, NAMES[cid])
    assert ctx.query("anchor") == 1
    assert ctx.run(sc) == exp and ctx.query("anchored_window") == 0 and ctx.query("anchor_sums") == 0   # default: not below 2^20 pairs
    ctx.set_option("anchor", 2)
    sums = 0
    for c in (0, 7, 11, 13, 16, 21):
        ctx.set_option("window_bits", c)
        assert ctx.run(sc) == exp, c
        t = ctx.last_timings()
        assert ctx.query("anchored_window") == BITS[cid] // t["window_bits"], c      # 1 + the last full window
        sums = max(sums, 1)
        assert ctx.query("anchor_sums") == sums       # ONE sum of bases [0, n), whatever the window size
        assert t["launches"] == 2                     # ... and the nested run that made it left no trace in the counters
    ctx.set_option("window_bits", 0)
    # carried chunks share the batch's anchored window; device scalars; a shorter run needs the sum of a shorter prefix
    for chunk in (n // 2 + 1, 1025):
        ctx.set_option("max_chunk", chunk)
        assert ctx.run(sc) == exp, chunk
        assert ctx.query("anchored_window") > 0 and ctx.last_timings()["launches"] == 2 * -(-n // chunk)
    assert ctx.run(torch.from_numpy(sc).cuda()) == exp
    ctx.set_option("max_chunk", 0)
    m = n - 1234
    assert ctx.run(np.ascontiguousarray(sc[:m]), npoints=m)[0] == oracle_msm_np(oracle, cid, bases[:m], np.ascontiguousarray(sc[:m]), m)
    assert ctx.query("anchor_sums") == 2
    assert ctx.run(sc) == exp and ctx.query("anchor_sums") == 2      # both sums are kept
    # anchor = 2 + the settings it does not combine with: plain digits, same bytes
    ctx.set_option("carry", 0)
    assert ctx.run(sc) == exp and ctx.query("anchored_window") == 0
    ctx.set_option("carry", 1)
    ctx.set_option("anchor", 0)
    assert ctx.run(sc) == exp and ctx.query("anchored_window") == 0
    # new bases: the sums of the old ones are gone
    ctx.set_option("anchor", 2)
    b2 = ea.generate_points(n, distinct=100, seed=77 + cid, curve=NAMES[cid])
    ctx.set_bases(b2)
    one = np.ascontiguousarray(sc[:n])
    assert ctx.run(one)[0] == oracle_msm_np(oracle, cid, b2, one, n) and ctx.query("anchor_sums") == 3
    ctx.close()


def test_all_zero_and_all_equal_scalars(ea, oracle):
    n = 5000
    bases = ea.generate_points(n, distinct=64, seed=2)
    ctx = ea.multi_scalar_mult_init(bases, "bls12_377_g1")
    ctx.set_option("anchor", 2)
    zero = np.zeros((n, 32), dtype=np.uint8)
    # every digit of the anchored window is -2^(c-1): one bucket holds -(sum of the bases), the host adds it back: infinity
    assert ctx.run(zero)[0] == oracle_msm_np(oracle, 0, bases, zero, n)
    same = np.tile(_scalars(0, 1, 9, special=False), (n, 1))
    assert ctx.run(same)[0] == oracle_msm_np(oracle, 0, bases, same, n)
    ctx.close()


@pytest.mark.parametrize("precompute,levels", [(1, 0), (1, 3)])
def test_anchored_window_with_tables_and_montgomery_scalars(ea, oracle, precompute, levels):
    n = 7000
    bases = ea.generate_points(n, distinct=300, seed=8)
    sc = _scalars(0, n, 12)
    exp = oracle_msm_np(oracle, 0, bases, sc, n)
    ctx = ea.MultiScalarMultContext("bls12_377_g1")
    ctx.set_option("precompute", precompute)
    ctx.set_option("table_levels", levels)
    ctx.set_option("anchor", 2)
    ctx.set_bases(bases)
    assert ctx.run(sc)[0] == exp and ctx.last_timings()["tables"] and ctx.query("anchored_window") > 0
    ctx.set_option("max_chunk", 2000)
    assert ctx.run(sc)[0] == exp
    ctx.close()
    # scalars handed over in Fr-Montgomery form: the nested sum-of-bases run uses plain ones all the same
    r = R[0]
    mont = np.zeros_like(sc)
    for i in range(n):
        v = int.from_bytes(sc[i].tobytes(), "little") % r
        mont[i] = np.frombuffer((v * (1 << 256) % r).to_bytes(32, "little"), dtype=np.uint8)
    canon = np.zeros_like(sc)
    for i in range(n):
        canon[i] = np.frombuffer((int.from_bytes(sc[i].tobytes(), "little") % r).to_bytes(32, "little"), dtype=np.uint8)
    ctx = ea.multi_scalar_mult_init(bases, "bls12_377_g1")
    ctx.set_option("anchor", 2)
    ctx.set_option("scalars_montgomery", 1)
    assert ctx.run(mont)[0] == oracle_msm_np(oracle, 0, bases, canon, n)
    assert ctx.query("anchored_window") > 0
    ctx.close()


def test_points_outside_the_subgroup(ea, oracle):
    """No assumption about the bases: points of order 2r (the assume_subgroup trick would be WRONG on them) give the oracle's bytes."""
    import random

    import pymodel as m

    C = m.BLS12_377_G1
    rng = random.Random(3)
    n = 4096
    T = (C.p - 1, 0)
    pts = [C.add(P, T) for P in m.random_points(C, 30, rng)] + m.random_points(C, 30, rng)
    bases = np.frombuffer(C.encode_affine_array([pts[i % 60] for i in range(n)]), dtype=np.uint8).reshape(n, 104).copy()
    sc = _scalars(0, n, 21)
    exp = oracle_msm_np(oracle, 0, bases, sc, n)
    for te in (1, 0):
        ctx = ea.MultiScalarMultContext("bls12_377_g1")
        ctx.set_option("twisted_edwards", te)
        ctx.set_option("anchor", 2)
        ctx.set_bases(bases)
        for c in (0, 12, 21):
            ctx.set_option("window_bits", c)
            assert ctx.run(sc)[0] == exp, (te, c)
            assert ctx.query("anchored_window") > 0
        ctx.close()


def test_default_setting_at_a_megapair(ea, oracle):
    """anchor = 1 (the default): from 2^20 pairs, where the window size has an anchored window worth >= 2 % of the additions.  The model
    decides the window size; whatever it picks, the bytes are the oracle's, and the same as with the option off."""
    n = (1 << 20) + 3
    for cid in (0, 1):
        bases = ea.generate_points(n, distinct=512, seed=40 + cid, curve=NAMES[cid])
        sc = _scalars(cid, n, 41 + cid)
        exp = oracle_msm_np(oracle, cid, bases, sc, n)
        ctx = ea.multi_scalar_mult_init(bases, NAMES[cid])
        assert ctx.run(sc)[0] == exp
        c_on, a_on = ctx.last_timings()["window_bits"], ctx.query("anchored_window")
        ctx.set_option("window_bits", 21 if cid == 0 else 17)     # 252 = 12 x 21; 255 = 15 x 17
        assert ctx.run(sc)[0] == exp and ctx.query("anchored_window") == (12 if cid == 0 else 15)
        ctx.set_option("window_bits", 20)                          # 253 = 12 x 20 + 13: nothing to gain, plain digits
        assert ctx.run(sc)[0] == exp and ctx.query("anchored_window") == 0
        ctx.set_option("window_bits", 0)
        ctx.set_option("anchor", 0)
        assert ctx.run(sc)[0] == exp and ctx.query("anchored_window") == 0
        print(f"{NAMES[cid]} n=2^20+3: anchor on -> c={c_on}, anchored window {a_on}; off -> c={ctx.last_timings()['window_bits']}")
        ctx.close()


def test_sharded_context_keeps_one_sum_per_shard(ea, oracle):
    """A sharded context (mi355_msm_create_sharded; here three logical shards on one device) anchors shard by shard: every shard holds the
    sum of ITS slice of the bases, computed in set_bases; the partial points fold to the oracle's bytes."""
    n = 20011
    for cid in (0, 1):
        bases = ea.generate_points(n, distinct=333, seed=90 + cid, curve=NAMES[cid])
        sc = _scalars(cid, n, 17 + cid)
        exp = oracle_msm_np(oracle, cid, bases, sc, n)
        ctx = ea.MultiScalarMultContext(NAMES[cid], devices=[0, 0, 0])
        ctx.set_option("anchor", 2)
        ctx.set_bases(bases)
        assert ctx.query("anchor_sums") == 3            # in set_bases: one per shard
        assert ctx.run(sc)[0] == exp and ctx.query("anchored_window") > 0 and ctx.query("anchor_sums") == 3
        ctx.set_option("force_peer_staging", 1)
        assert ctx.run(sc)[0] == exp
        ctx.close()
