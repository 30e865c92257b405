"""CPU: the 13 x 29 limb shape of BLS12-377 Fq without a GPU -- (1) the static worst-case proof of every 64-bit column of the
twisted-Edwards law on it (tools/limb_bounds29.py: positive margin everywhere, and the two refuted variants stay refuted),
(2) the HOST half of tests/test_gpu_limbs29.py: the same boundary records through g++'s build of the templates with the limb-bound
checker armed, against Python big integers and oracle/te_model.py, (3) the field product through the re-radixing steps the engine
uses at its boundaries (fe_28_to_29 / fe_29_to_28)."""
import ctypes
import os
import random
import subprocess
import sys

import pytest

import pymodel as m
import test_gpu_devtest as g
import test_gpu_limbs29 as t29
from conftest import ROOT


@pytest.fixture(scope="module")
def host_libs(built):
    return g.load_libs(False)


def test_static_column_bounds_hold():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "limb_bounds29.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "OVERFLOW" not in r.stdout
    lines = [ln for ln in r.stdout.splitlines() if "* 2^58 (k=" in ln]
    assert len(lines) >= 19                      # te_madd (3 + 4), te_add (5), te_add_quad (2 + 4), the re-radixing product
    margins = [float(ln.split(")")[-1].split("%")[0]) for ln in lines]
    assert min(margins) > 10.0, min(margins)     # the tightest column (Z3 = F' G) keeps 12.9 % of 2^64
    assert "no fixed point" in r.stdout          # R = 2^377 (13 reduction steps) stays refuted


@pytest.mark.parametrize("cls", t29.CLASSES, ids=[c[0] for c in t29.CLASSES])
def test_fe_mul_13x29_host(host_libs, cls):
    t29.test_fe_mul_13x29_at_the_operand_classes_of_the_law(host_libs, cls)


def test_law_13x29_host(host_libs):
    t29.test_twisted_edwards_law_13x29_matches_the_model(host_libs)
    t29.test_law_13x29_at_the_largest_class_m_limbs(host_libs)


def test_product_through_the_reradixing_steps(built):
    lib = ctypes.CDLL(os.path.join(ROOT, "2022-entries_amd", "libmsm_hosttest.so"))
    lib.ht_first_failure.restype = ctypes.c_char_p
    lib.ht_check_failures.restype = ctypes.c_long
    lib.ht_reset_checks()
    p = m.BLS12_377_G1.p
    rng = random.Random(29)
    out = ctypes.create_string_buffer(48)
    img = lambda v: ((v << 384) % p).to_bytes(48, "little")
    for i in range(3000):
        a, b = rng.randrange(p), rng.randrange(p)
        if i < 6:
            a, b = [(0, 5), (1, p - 1), (p - 1, p - 1), (p - 1, 1), (1 << 376, 1 << 376), ((1 << 377) % p, 3)][i]
        assert lib.ht_fe29_mul(img(a), img(b), out) == 0
        assert out.raw == img(a * b % p), i
    assert lib.ht_te29_extreme(8) == 0
    assert lib.ht_check_failures() == 0, lib.ht_first_failure()


def test_test_libraries_agree_on_the_shape_table(built):
    dev = ctypes.CDLL(os.path.join(ROOT, "2022-entries_amd", "libmsm_devtest.so"))
    host = g.load_libs(False)[1]
    for name, op in g.OPS.items():
        di, do, hi, ho = (ctypes.c_int() for _ in range(4))
        rd = dev.msm_devtest_shape(4, op, ctypes.byref(di), ctypes.byref(do))
        rh = host.ht_devop_shape(4, op, ctypes.byref(hi), ctypes.byref(ho))
        assert rd == rh and (di.value, do.value) == (hi.value, ho.value), name
        assert (rd == 0) == (name == "FE_MUL" or name.startswith("TE_")), name
