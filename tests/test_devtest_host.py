"""CPU: the HOST half of tests/test_gpu_devtest.py -- the same boundary records through the op table of csrc/devtest_ops.hpp as g++
compiles it (portable loops, limb-bound checker armed), checked against Python big integers and the affine model.  What this
proves here, without a GPU: the records respect every documented bound (no 64-bit column overflows, no biased subtraction
underflows), and the values are right.  The GPU test then requires the device build to return the same limbs."""
import ctypes
import os

import pytest

import test_gpu_devtest as g
from conftest import ROOT


@pytest.fixture(scope="module")
def host_libs(built):
    return g.load_libs(False)


@pytest.mark.parametrize("cid", [0, 1])
def test_field_ops_host(host_libs, cid):
    g.test_fe_mul_sqr_mul2_at_the_lazy_bounds(host_libs, cid)
    g.test_weak_reduce_and_bfi_step(host_libs, cid)


@pytest.mark.parametrize("cid", [2, 3])
def test_fp2_ops_host(host_libs, cid):
    g.test_fp2_products_in_every_operand_class(host_libs, cid)


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_xyzz_ops_host(host_libs, cid):
    g.test_xyzz_additions_match_the_affine_model(host_libs, cid)


def test_twisted_edwards_ops_host(host_libs):
    g.test_twisted_edwards_additions_match_the_model(host_libs)


def test_device_library_exports_and_shapes(built):
    """libmsm_devtest.so loads without a GPU and agrees with the host table on every record shape."""
    dev = ctypes.CDLL(os.path.join(ROOT, "2022-entries_amd", "libmsm_devtest.so"))
    host = g.load_libs(False)[1]
    assert hasattr(dev, "msm_devtest_run")
    for cid in (0, 1, 2, 3):
        for name, op in g.OPS.items():
            di, do, hi, ho = (ctypes.c_int() for _ in range(4))
            rd = dev.msm_devtest_shape(cid, op, ctypes.byref(di), ctypes.byref(do))
            rh = host.ht_devop_shape(cid, op, ctypes.byref(hi), ctypes.byref(ho))
            assert rd == rh and (di.value, do.value) == (hi.value, ho.value), (cid, name)
            if name.startswith("TE_"):
                assert (rd == 0) == (cid == 0)
