"""Build the native pieces in-tree (the .so files are git-ignored but travel with the gpurun snapshot).

  libmi355msm.so       hipcc --offload-arch=gfx950: kernels + engine + C ABI (the product)
  libmi355msm_debug.so the same with -DMSM_DEBUG on the engine and grouping units: device-side invariant checks (tests only)
  libmsm_hosttest.so   g++: the same fp28/curve templates for the host with the limb-bound checker (tests only)
  libmsm_devtest.so    hipcc: the same templates as element-wise test kernels (tests only)
  oracle/liboracle.so  gcc: CPU restatement of the arkworks algorithm (tests / smoke / bench cpu_baseline only)
  oracle/_ref/*        reference-derived cross-checks, only when /root/reference exists
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, **kw)


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MSM engine is HIP-only (no CPU fallback)")


ENGINE_UNITS = ["msm_engine.hip", "partition.hip", "kernels_377g1.hip", "kernels_381g1.hip", "kernels_377g2.hip", "kernels_381g2.hip", "kernels_377te.hip", "kernels_377g2p.hip", "kernels_381g2p.hip"]


def _depfile_deps(dfile: str):
    """Prerequisites listed by a make-style dependency file (hipcc -MD -MF), or None when it is missing / unreadable."""
    try:
        txt = open(dfile).read().replace("\\\n", " ")
    except OSError:
        return None
    deps = []
    for rule in txt.split("\n"):
        if ":" in rule:
            deps += [d for d in rule.split(":", 1)[1].split() if not d.startswith("/opt/") and not d.startswith("/usr/")]
    return deps or None


def build_engine(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950: one object per translation unit (the per-curve kernel units take minutes each --
    every field multiply is fully unrolled -- so they are compiled in parallel), then one link into libmi355msm.so.
    A unit is rebuilt when one of the files IT includes changed (dependency files written by hipcc -MD): a change to the host
    orchestration does not recompile the kernel units."""
    out = os.path.join(PKG, "libmi355msm.so")
    headers = [f for f in glob.glob(os.path.join(CSRC, "*")) if not f.endswith(".hip") and os.path.isfile(f)]
    abi_headers = glob.glob(os.path.join(ROOT, "include", "*.h"))   # only the engine unit includes the C ABI header
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    jobs, objs = [], []
    for unit in ENGINE_UNITS:
        src = os.path.join(CSRC, unit)
        obj = os.path.join(objdir, unit.replace(".hip", ".o"))
        dfile = obj + ".d"
        objs.append(obj)
        deps = _depfile_deps(dfile)
        if deps is None or not all(os.path.exists(d) for d in deps):
            deps = headers + [src] + (abi_headers if unit == "msm_engine.hip" else [])
        if force or _newer(obj, deps):
            cmd = [cc, "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-MD", "-MF", dfile, "-c", src, "-o", obj]
            print("+", " ".join(cmd), flush=True)
            jobs.append((unit, subprocess.Popen(cmd)))
    failed = [unit for unit, pr in jobs if pr.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(failed))
    if force or jobs or _newer(out, objs):
        _run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def _debug_engine_compile(force: bool = False):
    """Start the -DMSM_DEBUG compiles of the two units that differ from the product; returns (jobs, objs, out)."""
    out = os.path.join(PKG, "libmi355msm_debug.so")
    objdir = os.path.join(PKG, "build", "debug")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    headers = [f for f in glob.glob(os.path.join(CSRC, "*")) if not f.endswith(".hip") and os.path.isfile(f)]
    jobs, objs = [], []
    for unit in ENGINE_UNITS:
        if unit not in ("msm_engine.hip", "partition.hip"):
            objs.append(os.path.join(PKG, "build", unit.replace(".hip", ".o")))
            continue
        src = os.path.join(CSRC, unit)
        obj = os.path.join(objdir, unit.replace(".hip", ".o"))
        dfile = obj + ".d"
        objs.append(obj)
        deps = _depfile_deps(dfile)
        if deps is None or not all(os.path.exists(d) for d in deps):
            deps = headers + [src] + glob.glob(os.path.join(ROOT, "include", "*.h"))
        if force or _newer(obj, deps):
            cmd = [cc, "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-DMSM_DEBUG", "-MD", "-MF", dfile, "-c", src, "-o", obj]
            print("+", " ".join(cmd), flush=True)
            jobs.append((unit, subprocess.Popen(cmd)))
    return jobs, objs, out


def _debug_engine_link(jobs, objs, out, force: bool = False) -> str:
    failed = [unit for unit, pr in jobs if pr.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for (debug) " + ", ".join(failed))
    if force or jobs or _newer(out, objs):
        _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_debug_engine(force: bool = False) -> str:
    """libmi355msm_debug.so (tests only): the host orchestration and the grouping unit compiled with -DMSM_DEBUG -- invariant checks
    after every grouping level and after the accumulation (csrc/partition.hpp; the reference keeps such a self-check, disabled, in
    CMB Partition4096.cu:419-432) -- linked with the SAME kernel objects as the product, which are therefore brought up to date first
    (a stand-alone call on a clean tree, or after a kernel header changed, would otherwise fail to link or check stale kernels).
    tests/test_gpu_debug_build.py runs it."""
    dbg = _debug_engine_compile(force)
    build_engine(force)
    return _debug_engine_link(*dbg, force)


def build_hosttest(force: bool = False) -> str:
    out = os.path.join(PKG, "libmsm_hosttest.so")
    deps = glob.glob(os.path.join(CSRC, "*"))
    if force or _newer(out, deps):
        _run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, os.path.join(CSRC, "host_test_api.cpp")])
    return out


def build_devtest(force: bool = False) -> str:
    """libmsm_devtest.so (tests only): the arithmetic ops of csrc/devtest_ops.hpp as gfx950 kernels, one thread per raw limb record
    (tests/test_gpu_devtest.py compares them limb for limb with the host build of the same templates)."""
    out = os.path.join(PKG, "libmsm_devtest.so")
    deps = [f for f in glob.glob(os.path.join(CSRC, "*")) if os.path.isfile(f) and (f.endswith(".hpp") or f.endswith(".inc") or f.endswith("devtest.hip"))]
    if force or _newer(out, deps):
        _run([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared", "-o", out, os.path.join(CSRC, "devtest.hip")])
    return out


def build_shims(force: bool = False) -> list:
    """Harness-named symbol sets (include/mi355_msm_shims.h): tiny C objects linked against libmi355msm.so."""
    outs = []
    shim_dir = os.path.join(CSRC, "shims")
    variants = [("sppark_stateless.c", "sppark", ("377", "381")), ("zprize_harness.c", "zprize", ("377", "381")),
                ("yrrid_context.c", "yrrid", ("377",)), ("north_star_msm.c", "msm", ("377", "381"))]
    for src, name, curves in variants:
        for cv in curves:
            out = os.path.join(PKG, f"libmi355msm_{name}_{cv}.so")
            deps = [os.path.join(shim_dir, src)] + glob.glob(os.path.join(ROOT, "include", "*.h"))
            if force or _newer(out, deps):
                _run(["gcc", "-O2", "-shared", "-fPIC", f"-DFEATURE_BLS12_{cv}", "-o", out, os.path.join(shim_dir, src),
                      f"-L{PKG}", "-lmi355msm", "-Wl,-rpath,$ORIGIN"])
            outs.append(out)
    return outs


def build_oracle() -> str:
    _run(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return os.path.join(ROOT, "oracle", "liboracle.so")


def build_all(force: bool = False) -> None:
    dbg = _debug_engine_compile(force)   # (its two units compile while the product's do; it links the product's kernel objects)
    build_engine(force)
    _debug_engine_link(*dbg, force)
    build_hosttest(force)
    build_devtest(force)
    build_shims(force)
    build_oracle()


if __name__ == "__main__":
    build_all()
