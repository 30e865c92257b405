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


def _compile_kernel(instantiation):
    src = '#include "%s/2022-entries_amd/csrc/msm_kernels.cuh"\nnamespace msm {\n%s\n}\n' % (ROOT, instantiation)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "acc.hip"), "w").write(src)
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++20", "-c", "acc.hip", "-o", "acc.o", "-save-temps",
                            "-Rpass-analysis=kernel-resource-usage"], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        asm = open(os.path.join(d, "acc-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
        remarks = r.stderr
    body = asm[asm.index("_ZN3msm17k_accumulate_coop"):]
    body = body[:body.index("s_endpgm")]
    ops = re.findall(r"^\s+([a-z_0-9]+)", body, flags=re.M)
    blk = remarks[remarks.index("k_accumulate_coop"):]
    res = {k: int(re.search(pat, blk).group(1)) for k, pat in
           (("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"),
            ("vgprs", r"VGPRs: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"))}
    return body, ops, res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("law", ["sw", "te"])
def test_accumulate_kernel_isa(law):
    if law == "sw":
        inst = ("template __global__ void k_accumulate_coop<SwLaw<FpEl<Bls12_377_Fq>>>(const uint32_t*, const uint32_t*, uint32_t, uint32_t, "
                "uint32_t, const AffineDev*, SegOut, uint32_t, uint32_t*);")
    else:
        inst = ("template __global__ void k_accumulate_coop<TeLaw<Bls12_377_Fq>>(const uint32_t*, const uint32_t*, uint32_t, uint32_t, "
                "uint32_t, const TeAffineDev*, SegOut, uint32_t, uint32_t*);")
    body, ops, res = _compile_kernel(inst)
    assert "s_set_gpr_idx_on" not in body and "v_accvgpr" not in body
    assert body.count("scratch_") <= 8        # at most a couple of address registers parked outside the loop
    mads = ops.count("v_mad_u64_u32")
    carries = sum(ops.count(o) for o in ("v_addc_co_u32_e32", "v_addc_co_u32_e64", "v_addc_co_u32"))
    if law == "sw":
        # general add = 6 mul + 2 sqr + 1 fused dual product = 3542 MADs (3416 with the p0 = 1 shortcut); plus the rare doubling branch
        assert 3416 <= mads <= 8000, mads
    else:
        # 7 multiplications of 378 MADs, one straight-line body: no doubling / infinity branches at all
        assert 2646 <= mads <= 2700, mads
    assert carries < 50, carries            # the multiply-add chain is carry-free by construction
    assert res["scratch"] <= 32 and res["occupancy"] >= 2 and res["vgprs"] <= 256
    assert 256 * 128 <= res["lds"] <= 80 * 1024     # the record slots of the quad-cooperative gather; two blocks per CU fit in 160 KB
    # the gathers are 16-B per lane and the pieces cross lanes through LDS
    assert ops.count("ds_write_b128") >= 8 and ops.count("ds_read_b128") >= 7
