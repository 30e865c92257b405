"""GPU: arkworks' streaming accumulators over the engine -- ChunkedPippenger and HashMapPippenger
(ARK ec/src/msm/variable_base/stream_pippenger.rs:11-75, 78-140) -- against the CPU oracle on the concatenated input:
interleaved add patterns, buffer sizes that divide the input or not, duplicate bases (summed modulo r before the MSM),
scalars that cancel, an empty finalize, reuse after finalize, Fr-Montgomery scalars, BLS12-381 and G2."""
import ctypes

import numpy as np
import pytest

import pymodel as m
from conftest import oracle_msm_np

pytestmark = pytest.mark.gpu

R377_TOP = 0x12ab655e9a2ca556


def _scalars(n, seed, top=R377_TOP):
    rng = np.random.default_rng(seed)
    limbs = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] %= np.uint64(top)
    return limbs.view(np.uint8).reshape(n, 32)


def _oracle(oracle, cid, curve, ea, bases, sc):
    out = ctypes.create_string_buffer(ea.projective_bytes(curve))
    assert oracle.oracle_msm(cid, bases.ctypes.data, ea.affine_stride(curve), sc.ctypes.data, len(sc), out, 0) == 0
    return out.raw


@pytest.mark.parametrize("buf", [1, 7, 1000, 5003, 100000])
def test_chunked_pippenger_matches_one_msm(ea, oracle, buf):
    n = 5003
    bases = ea.generate_points(n, distinct=700, seed=1)
    sc = _scalars(n, 2)
    sc[11] = 0
    exp = oracle_msm_np(oracle, 0, bases, sc, n)
    cp = ea.ChunkedPippenger(buf)
    rng = np.random.default_rng(buf)
    pos = 0
    while pos < n:                      # ragged pieces: one pair at a time, a few, many
        step = int(rng.choice([1, 1, 2, 17, 300, 2500]))
        cp.add(bases[pos:pos + step], sc[pos:pos + step])
        pos += step
    assert cp.query("buffered") == n % buf and cp.query("flushes") == n // buf
    assert cp.finalize() == exp
    # finalize left it empty: an empty finalize is the point at infinity, and the object can be reused
    inf = cp.finalize()
    assert inf[96:] == bytes(48) and cp.query("buffered") == 0
    cp.add(bases[:100], sc[:100])
    assert cp.finalize() == oracle_msm_np(oracle, 0, bases, sc, 100)
    cp.close()


def test_hashmap_pippenger_merges_equal_bases(ea, oracle):
    """2000 pairs over 37 distinct bases: the map never holds more than 37 entries, so with buf_size 64 nothing is flushed before
    finalize and the single MSM has 37 pairs; with buf_size 16 the map is flushed every 16 distinct bases.  Either way the
    result is the MSM of all 2000 pairs."""
    c = m.BLS12_377_G1
    n, distinct = 2000, 37
    pool = ea.generate_points(distinct, distinct=distinct, seed=9)
    rng = np.random.default_rng(3)
    pick = rng.integers(0, distinct, size=n)
    bases = np.ascontiguousarray(pool[pick])
    sc = _scalars(n, 4)
    # r - k and k on the same base: the entry sums to zero modulo r
    k = int.from_bytes(sc[5].tobytes(), "little")
    sc[6] = np.frombuffer(((c.r - k) % c.r).to_bytes(32, "little"), dtype=np.uint8)
    bases[6] = bases[5]
    exp = oracle_msm_np(oracle, 0, bases, sc, n)
    for buf, flushes in ((64, 0), (16, None), (1, None)):
        hp = ea.HashMapPippenger(buf)
        for lo in range(0, n, 333):
            hp.add(bases[lo:lo + 333], sc[lo:lo + 333])
        if flushes is not None:
            assert hp.query("flushes") == flushes and hp.query("buffered") == distinct and hp.query("merged") == n - distinct
        assert hp.finalize() == exp, buf
        hp.close()
    # a base at infinity is a key like any other (flag byte set): its scalars merge and contribute nothing
    bases[40:44, 96] = 1
    exp = oracle_msm_np(oracle, 0, bases, sc, n)
    hp = ea.HashMapPippenger(50)
    hp.add(bases, sc)
    assert hp.finalize() == exp
    hp.close()


def test_hashmap_pippenger_fr_montgomery_scalars(ea, oracle):
    """arkworks hands HashMapPippenger Fr values, i.e. Montgomery images a * 2^256 mod r: sums of images are images of sums, and the
    flush converts on the device ("scalars_montgomery")."""
    c = m.BLS12_377_G1
    n, distinct = 600, 50
    pool = ea.generate_points(distinct, distinct=distinct, seed=19)
    rng = np.random.default_rng(5)
    bases = np.ascontiguousarray(pool[rng.integers(0, distinct, size=n)])
    ks = [int(rng.integers(0, 1 << 62)) ** 4 % c.r for _ in range(n)]
    plain = np.frombuffer(b"".join(k.to_bytes(32, "little") for k in ks), dtype=np.uint8).reshape(n, 32)
    mont = np.frombuffer(b"".join((k * (1 << 256) % c.r).to_bytes(32, "little") for k in ks), dtype=np.uint8).reshape(n, 32)
    exp = oracle_msm_np(oracle, 0, bases, np.ascontiguousarray(plain), n)
    hp = ea.HashMapPippenger(20)
    hp.set_option("scalars_montgomery", 1)
    hp.add(bases, np.ascontiguousarray(mont))
    assert hp.query("flushes") >= 2
    assert hp.finalize() == exp
    hp.close()


@pytest.mark.parametrize("curve,cid", [("bls12_381_g1", 1), ("bls12_377_g2", 2), ("bls12_381_g2", 3)])
def test_stream_other_curves(ea, oracle, curve, cid):
    n = 900
    bases = ea.generate_points(n, distinct=60, seed=6, curve=curve)
    sc = _scalars(n, 8, top=0x12ab655e9a2ca556)
    exp = _oracle(oracle, cid, curve, ea, bases, sc)
    cp = ea.ChunkedPippenger(256, curve)
    cp.add(bases, sc)
    assert cp.query("flushes") == 3 and cp.finalize() == exp
    cp.close()
    hp = ea.HashMapPippenger(64, curve)           # more room than distinct bases: every repeat merges, one MSM of 60 pairs
    hp.add(bases, sc)
    assert hp.query("merged") == n - 60 and hp.query("flushes") == 0 and hp.finalize() == exp
    hp.close()
    hp = ea.HashMapPippenger(32, curve)           # the bases cycle with period 60: no repeat inside any 32 distinct ones
    hp.add(bases, sc)
    assert hp.query("merged") == 0 and hp.query("flushes") == n // 32 and hp.finalize() == exp
    hp.close()


def test_stream_argument_errors(ea):
    with pytest.raises(ea.MsmError):
        ea.ChunkedPippenger(0)
    cp = ea.ChunkedPippenger(10)
    with pytest.raises(ValueError):
        cp.add(np.zeros((3, 104), dtype=np.uint8), np.zeros((2, 32), dtype=np.uint8))
    with pytest.raises(ea.MsmError):
        cp.set_option("no_such_option", 1)
    cp.close()
