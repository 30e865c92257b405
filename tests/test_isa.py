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


def _compile_kernel(instantiation, mangled="_ZN3msm17k_accumulate_glds"):
    src = '#include "%s/2022-entries_amd/csrc/msm_kernels.hpp"\nnamespace msm {\n%s\n}\n' % (ROOT, instantiation)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "acc.hip"), "w").write(src)
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++20", "-c", "acc.hip", "-o", "acc.o", "-save-temps",
                            "-Rpass-analysis=kernel-resource-usage"], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        asm = open(os.path.join(d, "acc-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
        remarks = r.stderr
    body = asm[asm.index(mangled):]
    body = body[:body.index("s_endpgm")]
    ops = re.findall(r"^\s+([a-z_0-9]+)", body, flags=re.M)
    blk = remarks[remarks.index(mangled.split("msm")[-1].lstrip("0123456789")):]
    res = {k: int(re.search(pat, blk).group(1)) for k, pat in
           (("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"),
            ("vgprs", r"VGPRs: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"))}
    return body, ops, res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("law", ["sw", "te", "te28"])
def test_accumulate_kernel_isa(law):
    """law = te: the product path of BLS12-377 G1 since round 6 (twisted Edwards on 13 x 29 limbs, csrc/fp28.hpp); te28: the same law on
    14 x 28 limbs (rounds 2-5, still instantiable: -DMSM_TE_LIMBS29=0), whose pins stay so that the generalised fe_mul provably emits
    the code it always did."""
    if law == "sw":
        inst = ("template __global__ void k_accumulate_glds<SwLaw<FpEl<Bls12_377_Fq>>>(const uint2*, const uint32_t*, uint32_t, "
                "const AffineDev*, SegOut, uint32_t, uint32_t*);")
    else:
        inst = ("template __global__ void k_accumulate_glds<TeLaw<%s>>(const uint2*, const uint32_t*, uint32_t, "
                "const TeAffineDev*, SegOut, uint32_t, uint32_t*);" % ("Bls12_377_Fq29" if law == "te" else "Bls12_377_Fq"))
    body, ops, res = _compile_kernel(inst)
    assert "s_set_gpr_idx_on" not in body and "v_accvgpr" not in body
    assert body.count("scratch_") <= 8        # at most a couple of address registers parked outside the loop
    mads = ops.count("v_mad_u64_u32")
    carries = sum(ops.count(o) for o in ("v_addc_co_u32_e32", "v_addc_co_u32_e64", "v_addc_co_u32"))
    if law == "sw":
        # general add = 6 mul + 2 sqr + 1 fused dual product = 3542 MADs (3416 with the p0 = 1 shortcut); plus the rare doubling branch
        assert 3416 <= mads <= 8000, mads
    elif law == "te":
        # 7 multiplications of 169 + 14 * 12 = 337 MADs (13 limbs, 14 Montgomery steps), one straight-line body
        assert 2359 <= mads <= 2410, mads
    else:
        # 7 multiplications of 378 MADs, one straight-line body: no doubling / infinity branches at all
        assert 2646 <= mads <= 2700, mads
    assert carries < 50, carries            # the multiply-add chain is carry-free by construction
    assert res["scratch"] <= 32 and res["occupancy"] >= 2 and res["vgprs"] <= 256
    assert 32 * 1024 <= res["lds"] <= 53 * 1024     # 4 waves x 4 records x SECT sectors x 1 KB (+ skew): three blocks per CU fit in 160 KB
    # the gathers are LDS-DMA (global -> LDS, no VGPR staging, no ds_write): 4 records x SECT sectors per addition
    sect = 2 if law == "sw" else 3
    assert ops.count("global_load_lds_dwordx4") == 2 * 4 * sect and ops.count("ds_write_b128") == 0
    # the sorted entries arrive through a register queue refilled by back-to-back 16-byte loads (a whole 64-B sector for the
    # twisted-Edwards kernel, half a sector for XYZZ): no 8-byte entry load per iteration is left
    eq = 4 if law.startswith("te") else 2
    assert ops.count("global_load_dwordx2") == 0 and ops.count("global_load_dwordx4") >= 2 * eq, (ops.count("global_load_dwordx2"), ops.count("global_load_dwordx4"))
    if law.startswith("te"):
        # Y - X and Y + X are read from each other's sector for a negated base: per-lane LDS addresses, 13 / 14 selects left (2dXY)
        assert ops.count("v_cndmask_b32_e64") <= 24 and res["vgprs"] <= 168      # 3 waves/SIMD resident
        # p = 1 mod 2^28 (and mod 2^29): the Montgomery step of a low column is v_lshl_add_u64 + v_bfi_b32 + v_lshrrev_b64 (fp28.hpp
        # MSM_MONT_STEP) -- 14 per multiplication in either limb shape, 7 multiplications; the rest of the VALU stream is bounded below
        assert ops.count("v_bfi_b32") == 98, ops.count("v_bfi_b32")
        valu = [o for o in ops if o.startswith("v_")]
        # whole kernel (static count), prologue, flushes and the queue rotation (14 moves) included.  13 x 29: two carry passes per
        # addition (te_tail: F and H) are part of it -- 3107 VALU per trip against 3348 (tools/isa_histogram.py)
        # + ~140 in the cold block of carried batches (SegOutT::carry_in, msm_kernels.hpp carry_begin_run: a run of a later chunk starts from
        # the stored bucket), outside the per-addition path -- tools/isa_histogram.py te29 separates the two: 3107 per trip, 223 cold
        assert len(valu) - mads <= 1200, len(valu) - mads
        if law == "te":
            assert len(valu) <= 3650, len(valu)
    else:
        # the common path of the mixed addition keeps neither base coordinate alive (curve.hpp xyzz_madd_common): 3 waves/SIMD
        assert res["vgprs"] <= 168, res["vgprs"]
    # selects are emitted in the VOP3 form: back-to-back v_cndmask_b32_e32 (mask implicit in VCC) issue at 22.9 cycles in isolation
    # against 4.2 for v_cndmask_b32_e64 (profiles/r02_ubench_valu_w4.txt).  In this kernel the difference did not show
    # (profiles/r02_ab_cndmask.txt); the form is pinned anyway so that a scheduling change cannot bring the slow case back
    # (static count of the whole kernel: the cold carry-in block of the Edwards kernel holds one more)
    assert ops.count("v_cndmask_b32_e32") <= 3, ops.count("v_cndmask_b32_e32")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("nb", [5, 1])
def test_paired_g2_accumulate_kernel_isa(nb):
    """G2 with every Fp2 value on two lanes (csrc/fp2pair.hpp, SwPairLaw): the point of the form is TWO waves per SIMD, so the
    kernel must fit 256 registers with NO scratch (the one-lane form needs 356 and spilled 138-412 VGPRs at two waves:
    profiles/r02_ab_g2_waves.txt), partner limbs must travel by DPP quad permutes (no LDS round trip, no ds_bpermute), and an Fp2
    product must stay ONE fused dual product per lane: 10 per mixed addition (8M + 2S), 574 (p0 = 1) or 588 multiply-adds each."""
    fq = "Bls12_377_Fq" if nb == 5 else "Bls12_381_Fq"
    inst = ("template __global__ void k_accumulate_glds<SwPairLaw<%s, %d>>(const uint2*, const uint32_t*, uint32_t, "
            "const AffineDevT<Fe2>*, SegOutT<Fe2>, uint32_t, uint32_t*);" % (fq, nb))
    body, ops, res = _compile_kernel(inst)
    assert res["scratch"] == 0 and res["occupancy"] >= 2 and res["vgprs"] <= 256, res
    assert "s_set_gpr_idx_on" not in body and "v_accvgpr" not in body and "scratch_" not in body
    assert "ds_bpermute" not in body and "ds_swizzle" not in body
    per_mul = 574 if nb == 5 else 588
    mads = ops.count("v_mad_u64_u32")
    # the common path (10 products) plus the rare same-x branch (doubling of an affine point: 9 more), statically
    assert 10 * per_mul <= mads <= 20 * per_mul, mads
    dpp = body.count("quad_perm")
    assert 6 * 42 <= dpp <= 20 * 42, dpp          # at most three permutes per limb and product; shared operands are permuted once
    # a quad gathers the records of its TWO pairs: 2 records x 4 sectors by LDS-DMA, at the prologue and in the loop
    assert ops.count("global_load_lds_dwordx4") == 2 * 2 * 4 and ops.count("ds_write_b128") == 0
    assert 4 * (2 * 4 * 1024) <= res["lds"] <= 40 * 1024, res["lds"]     # 8 KB per wave (+ skew): two blocks per CU and room to spare
    assert ops.count("v_cndmask_b32_e32") <= 20, ops.count("v_cndmask_b32_e32")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_quad_addition_kernel_isa():
    """The latency form of the scan step: four lanes per addition (te.hpp te_add_quad).  Three multiplications per lane (the
    one-lane unified addition has nine), operands exchanged by DPP quad permutes, small enough for 4 waves/SIMD, no scratch."""
    inst = ("template __global__ void k_reduce_scan_step_quad<TeQuad<Bls12_377_Fq29>>(const XyzzDev*, const XyzzDev*, XyzzDev*, uint32_t, uint32_t, "
            "uint32_t, uint32_t, uint32_t*);")
    body, ops, res = _compile_kernel(inst, "_ZN3msm23k_reduce_scan_step_quad")
    mads = ops.count("v_mad_u64_u32")
    assert 3 * 337 <= mads <= 3 * 337 + 30, mads   # 13 x 29 limbs (csrc/fp28.hpp)
    assert res["scratch"] == 0 and res["vgprs"] <= 128, res
    assert body.count("quad_perm") >= 6 * 13       # X<->Y swap of both operands, A / B / Z1Z2 / C to every lane
    assert "ds_bpermute" not in body and "ds_swizzle" not in body


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_partition_kernels_isa():
    """The bucket-grouping kernels (csrc/partition.hpp): no scratch, the tile's scalars live in registers (<= 128 VGPRs at 1024
    threads per block), LDS staging fits one block per CU with room to spare, and no library sort is linked."""
    src = '#include "%s/2022-entries_amd/csrc/partition.hip"\n' % ROOT
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "part.hip"), "w").write(src)
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++20", "-c", "part.hip", "-o", "part.o",
                            "-Rpass-analysis=kernel-resource-usage"], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        remarks = r.stderr
    seen = 0
    for blk in remarks.split("Function Name: ")[1:]:
        name = blk.split()[0]
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk).group(1))
        vgprs = int(re.search(r"VGPRs: (\d+)", blk).group(1))
        lds = int(re.search(r"LDS Size \[bytes/block\]: (\d+)", blk).group(1))
        assert scratch == 0, (name, scratch)
        if "k_l1_scatter" in name or "k_pass_scatter" in name:
            assert vgprs <= 128 and 64 * 1024 <= lds <= 150 * 1024, (name, vgprs, lds)
            seen += 1
    assert seen >= 5     # 4 instantiations of the level-1 scatter + the generic pass
    blob = open(os.path.join(ROOT, "2022-entries_amd", "libmi355msm.so"), "rb").read()
    assert b"rocprim" not in blob and b"k_l1_scatter" in blob and b"k_pass_scatter" in blob
