#!/usr/bin/env python3
"""Opcode classes of the accumulation loop, per mixed addition, on the cross-compiled gfx950 ISA (no GPU needed).

VERDICT r5 item 3: the accumulate kernels are VALU-issue-bound (VALUBusy 0.96), so the only term the code controls is the number of
VALU instructions that are NOT multiply-adds.  This tool compiles k_accumulate_glds for one group-law policy with hipcc -save-temps,
finds the loop (the backward branch whose body holds the most v_mad_u64_u32) and prints what one trip -- one mixed addition -- issues,
by class.  The static count of the loop body is exact for the twisted-Edwards law (one straight-line body); for the short-Weierstrass
laws the body also contains the rare same-x branch, which is reported separately (blocks reached only through a forward branch over
them are listed as `cold`).

    python tools/isa_histogram.py                 # the three product kernels + the 14 x 28 Edwards law for comparison
    python tools/isa_histogram.py te29 --json     # one law, machine-readable
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

LAWS = {
    "te29": ("TeLaw<Bls12_377_Fq29>", "const TeAffineDev*, SegOut", "BLS12-377 G1, twisted Edwards, 13 x 29 limbs (the product path)"),
    "te28": ("TeLaw<Bls12_377_Fq>", "const TeAffineDev*, SegOut", "BLS12-377 G1, twisted Edwards, 14 x 28 limbs (rounds 2-5)"),
    "sw381": ("SwLaw<FpEl<Bls12_381_Fq>>", "const AffineDev*, SegOut", "BLS12-381 G1, XYZZ"),
    "sw377": ("SwLaw<FpEl<Bls12_377_Fq>>", "const AffineDev*, SegOut", "BLS12-377 G1, XYZZ (the fallback path)"),
    "g2p377": ("SwPairLaw<Bls12_377_Fq, 5>", "const AffineDevT<Fe2>*, SegOutT<Fe2>", "BLS12-377 G2, XYZZ, Fp2 on two lanes"),
    "g2p381": ("SwPairLaw<Bls12_381_Fq, 1>", "const AffineDevT<Fe2>*, SegOutT<Fe2>", "BLS12-381 G2, XYZZ, Fp2 on two lanes"),
}

CLASSES = [
    ("mad", lambda o: o == "v_mad_u64_u32"),
    ("shift", lambda o: re.match(r"v_(lshl|lshr|ashr)", o) is not None and not o.startswith("v_lshl_add_u64")),
    ("add64", lambda o: o.startswith("v_lshl_add_u64")),
    ("logic", lambda o: re.match(r"v_(and|or|xor|not|bfi|bfe|and_or|or3)", o) is not None),
    ("addsub", lambda o: re.match(r"v_(add|sub|subrev|add3|addc|subb)", o) is not None),
    ("mul_lo", lambda o: re.match(r"v_mul_(lo|hi|u32|i32)", o) is not None),
    ("select", lambda o: o.startswith("v_cndmask")),
    ("cmp", lambda o: o.startswith("v_cmp")),
    ("mov", lambda o: re.match(r"v_(mov|readlane|readfirstlane|writelane|accvgpr|swap)", o) is not None),
    ("valu_other", lambda o: o.startswith("v_")),
    ("lds", lambda o: o.startswith("ds_")),
    ("vmem", lambda o: re.match(r"(global|buffer|flat|scratch)_", o) is not None),
    ("salu", lambda o: o.startswith("s_") and not o.startswith("s_waitcnt") and not o.startswith("s_nop")),
    ("wait", lambda o: o.startswith("s_waitcnt") or o.startswith("s_nop")),
]


def classify(op):
    for name, pred in CLASSES:
        if pred(op):
            return name
    return "other"


def compile_kernel(law, extra_flags=()):
    tmpl, args, _ = LAWS[law]
    inst = "template __global__ void k_accumulate_glds<%s>(const uint2*, const uint32_t*, uint32_t, %s, uint32_t, uint32_t*);" % (tmpl, args)
    src = '#include "%s/2022-entries_amd/csrc/msm_kernels.hpp"\nnamespace msm {\n%s\n}\n' % (ROOT, inst)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "acc.hip"), "w").write(src)
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++20", "-c", "acc.hip", "-o", "acc.o", "-save-temps",
                            "-Rpass-analysis=kernel-resource-usage", *extra_flags], cwd=d, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-3000:])
        asm = open(os.path.join(d, "acc-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    body = asm[asm.index("_ZN3msm17k_accumulate_glds"):]
    body = body[:body.index("s_endpgm")]
    res = {}
    for k, pat in (("vgprs", r"VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, r.stderr)
        res[k] = int(m.group(1)) if m else None
    return body, res


def loop_histogram(body):
    """-> (per-class counts of the hot loop, counts of cold blocks inside its address range, total static counts)"""
    lines = body.split("\n")
    labels, instrs = {}, []   # label -> instruction index; instrs = [(op, operand text)]
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", ln)
        if m and not ln.lstrip().startswith((".", ";")):
            instrs.append((m.group(1), m.group(2)))
    # backward branches = loops; take the one with the most multiply-adds inside
    best = None
    for i, (op, arg) in enumerate(instrs):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = arg.split()[0].rstrip(",") if arg else ""
            if tgt in labels and labels[tgt] <= i:
                lo, hi = labels[tgt], i + 1
                mads = sum(1 for o, _ in instrs[lo:hi] if o == "v_mad_u64_u32")
                if best is None or mads > best[0]:
                    best = (mads, lo, hi)
    if best is None:
        raise RuntimeError("no loop found")
    _, lo, hi = best
    # cold blocks.  Forward conditional branches that skip a range holding multiply-adds delimit the whole addition under
    # `if (add_now)`, the general case under `if (!fresh)` ... -- ranges that hold (nearly) all the multiply-adds of the range around them:
    # HOT -- and the exceptional paths: the same-x branch of the short-Weierstrass laws (doubling / cancellation, taken ~never on random
    # input; somewhat fewer multiply-adds than the general addition next to it) and the bucket flush at a change of key (its
    # multiply-adds are address arithmetic): COLD.  Rule: a skipped range is cold iff it holds less than half of the multiply-adds of
    # the smallest skipped range around it (or of the loop).
    def n_mad(a, b):
        return sum(1 for j in range(a, b) if instrs[j][0] == "v_mad_u64_u32")

    spans = []
    for i in range(lo, hi):
        op, arg = instrs[i]
        if op.startswith("s_cbranch"):
            tgt = arg.split()[0].rstrip(",")
            if tgt in labels and i < labels[tgt] <= hi:
                a, b = i + 1, labels[tgt]
                if n_mad(a, b):
                    spans.append((a, b))
    cold = [False] * len(instrs)
    for a, b in spans:
        parents = [(a2, b2) for a2, b2 in spans if a2 <= a and b <= b2 and (a2, b2) != (a, b)]
        pa, pb = min(parents, key=lambda x: x[1] - x[0]) if parents else (lo, hi)
        if 2 * n_mad(a, b) < n_mad(pa, pb):
            for j in range(a, b):
                cold[j] = True
    hot, cold_h, total = {}, {}, {}
    for i, (op, _) in enumerate(instrs):
        c = classify(op)
        total[c] = total.get(c, 0) + 1
        if lo <= i < hi:
            d = cold_h if cold[i] else hot
            d[c] = d.get(c, 0) + 1
    return hot, cold_h, total, (lo, hi, len(instrs))


def report(law, as_json=False, extra_flags=()):
    body, res = compile_kernel(law, extra_flags)
    hot, cold, total, (lo, hi, n) = loop_histogram(body)
    valu = sum(v for k, v in hot.items() if k in ("mad", "shift", "add64", "logic", "addsub", "mul_lo", "select", "cmp", "mov", "valu_other"))
    out = {"law": law, "what": LAWS[law][2], "resources": res, "loop_instructions": hi - lo, "static_instructions": n, "hot": hot, "cold": cold,
           "valu_per_addition": valu, "mad_per_addition": hot.get("mad", 0), "non_mad_valu": valu - hot.get("mad", 0),
           "mad_share": round(hot.get("mad", 0) / valu, 4) if valu else None}
    if as_json:
        return out
    print("%s -- %s" % (law, LAWS[law][2]))
    print("  VGPRs %s, scratch %s B/lane, %s waves/SIMD; loop = %d of %d static instructions" % (res["vgprs"], res["scratch"], res["occupancy"], hi - lo, n))
    print("  per trip (one mixed addition), hot path:   VALU %d = MAD %d + other %d   (MAD share %.3f)" % (valu, out["mad_per_addition"], out["non_mad_valu"], out["mad_share"]))
    print("    " + "  ".join("%s %d" % (k, hot[k]) for k, _ in CLASSES if k in hot))
    if cold:
        cv = sum(v for k, v in cold.items() if k not in ("lds", "vmem", "salu", "wait", "other"))
        print("  cold blocks inside the loop (same-x branch of the XYZZ laws, bucket flush):  VALU %d, MAD %d" % (cv, cold.get("mad", 0)))
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    as_json = "--json" in sys.argv
    laws = args or ["te29", "te28", "sw381", "sw377", "g2p377", "g2p381"]
    outs = [report(l, as_json) for l in laws]
    if as_json:
        print(json.dumps(outs, indent=1))
