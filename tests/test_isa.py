"""CPU: properties of the generated gfx950 code for the hot kernel, checked on the cross-compiled ISA.

The field arithmetic relies on hipcc fully unrolling the limb loops; a partially unrolled loop silently turns into dynamic
register indexing (s_set_gpr_idx_on / scratch) and costs an order of magnitude.  This test pins what DESIGN.md claims:
no scratch, no dynamic indexing, 2 waves/SIMD, and a carry-free multiply-add chain."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_accumulate_kernel_isa():
    src = '#include "%s/2022-entries_amd/csrc/msm_kernels.cuh"\nnamespace msm {\n' \
          'template __global__ void k_accumulate<SwLaw<FpEl<Bls12_377_Fq>>>(const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t, ' \
          'const AffineDev*, SegOut, uint32_t, uint32_t*);\n}\n' % ROOT
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "acc.hip"), "w").write(src)
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++20", "-c", "acc.hip", "-o", "acc.o", "-save-temps",
                            "-Rpass-analysis=kernel-resource-usage"], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        asm = open(os.path.join(d, "acc-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
        remarks = r.stderr
    body = asm[asm.index("_ZN3msm12k_accumulate"):]
    body = body[:body.index("s_endpgm")]
    assert "s_set_gpr_idx_on" not in body and "v_accvgpr" not in body
    assert body.count("scratch_") <= 8        # at most a couple of address registers parked outside the loop
    ops = re.findall(r"^\s+([a-z_0-9]+)", body, flags=re.M)
    mads = ops.count("v_mad_u64_u32")
    carries = sum(ops.count(o) for o in ("v_addc_co_u32_e32", "v_addc_co_u32_e64", "v_addc_co_u32"))
    # general add = 6 mul (392) + 2 sqr (301) + 1 fused dual product (588); plus the rare doubling branch
    assert 3542 <= mads <= 8000, mads
    assert carries < 50, carries            # the multiply-add chain is carry-free by construction
    blk = remarks[remarks.index("k_accumulate"):]
    assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk).group(1)) <= 32
    assert int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", blk).group(1)) >= 2
    assert int(re.search(r"VGPRs: (\d+)", blk).group(1)) <= 256
