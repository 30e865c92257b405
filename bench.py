#!/usr/bin/env python3
"""bench.py -- BLS12-377 G1 MSM throughput on MI355X (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps 5 --warmup 1
  python bench.py --only headline                       (the timed loop and its re-takes only: seconds, not minutes)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one MSM of 2^npow point-scalar pairs per GPU (default 2^26, BASELINE.json configs[1]) with the bases
AND the scalars already resident in HBM: device scalars -> digits -> sort -> bucket accumulation -> bucket reduction ->
window sums to the host -> Horner fold; with N > 1 ranks each rank owns a disjoint slice, the N 144-byte partials are
all-gathered with RCCL and folded on every rank.  Rank 0 prints ONE JSON line.

Workload per N: N = 1, 2, 4 -> 2^26 pairs per GPU (weak scaling from BASELINE configs[1]); N = 8 -> BASELINE configs[3], the
2^28-pair MSM sharded 8 ways (2^25 per GPU), with the 2^26-per-GPU weak-scaling point measured in the same run as a
secondary object.  `--total-npow T` fixes the GLOBAL size for any N (per GPU: 2^T / N).

`value` is EXACTLY --steps steps after --warmup, barrier + synchronize on both sides, max over ranks.  Because the kernel is
power-limited and boxes differ by +-3 %, the same K-step loop is then re-taken `--repeat` more times (`headline_samples`: median /
min / max) and the GPU's shader clock and socket power are sampled by a thread DURING every loop (`config.clock_MHz_*`,
`config.power_W_*`): a line carries what explains its own variance.

`roofline` is for the dominant kernel (bucket accumulation, k_accumulate_glds): algorithmic bytes = 128 B/pair
(32 B scalar + 96 B affine base) x pairs per launch, over its HIP-event duration on the launch stream; `peak` is the
guide's 8 TB/s, `peak_measured` a device-to-device copy probe of this very box.
`cpu_baseline` times oracle/liboracle.so -- the C restatement of arkworks' VariableBaseMSM, one thread per window like
rayon -- on a bounded sample of the same workload, rank 0, N = 1 only; when a cargo toolchain is on the box, real ark-ec is run
beside it through rust/benches/msm.rs (`cpu_baseline.ark_ec`).  It is a reported baseline, not the target.

Layout: the secondary measurements of the N = 1 line are functions that CREATE AND CLOSE THEIR OWN contexts (`measure_*`); the
only ones that borrow the headline context run before its single close() in main().
"""
import argparse
import ctypes
import glob
import json
import os
import shutil
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails (hipIpcGetMemHandle)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBPS = 6300.0    # MI355X_MICROARCH.md: what a streaming kernel reaches of the 8 TB/s
BYTES_PER_PAIR = {0: 128.0, 1: 128.0, 2: 224.0, 3: 224.0}   # 32 B scalar + 96 B affine base (G2: + 192 B)
R377_TOP = 0x12ab655e9a2ca556   # top 64-bit limb of the BLS12-377 scalar modulus (ARKC bls12_377/src/fields/fr.rs:24)
R381_TOP = 0x73eda753299d7d48
REF_CLOCK_GHZ = 1.95            # the clock the accumulate kernel ran at on the boxes of rounds 2-4 (profiles/r04_pmc_k_accumulate.json)
PMC_ROUND = "r06"
STAGES = ("digits", "sort", "accumulate", "segreduce", "bucket_reduce", "total")


MAD_PEAK = 1024 * 64 / 4.3 * 2.4e9   # lane-level v_mad_u64_u32 per second: 1024 SIMDs x 64 lanes / 4.3 cycles at the nominal 2.4 GHz (profiles/r01_ubench_valu_*.txt)


def r_top(cid):
    return R381_TOP if cid in (1, 3) else R377_TOP


def uniform_scalars(n, top_limb, device, seed):
    """n integers uniform on [0, top_limb * 2^192), a subset of [0, r): 4 x u64 little-endian limbs as a uint8 tensor."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    limbs = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=device, generator=g)
    bits = top_limb.bit_length()
    top = limbs[:, 3] & ((1 << bits) - 1)
    for _ in range(64):
        # (torch.where, not boolean indexing: masked assignment drags torch's own rocPRIM partition kernels into the trace)
        bad = top >= top_limb
        if int(bad.sum().item()) == 0:
            break
        fresh = torch.randint(0, min(1 << bits, (1 << 63) - 1), (n,), dtype=torch.int64, device=device, generator=g)
        top = torch.where(bad, fresh, top)
    limbs[:, 3] = top
    return limbs.view(torch.uint8).reshape(n, 32)


def kernel_source_sha16():
    """Identity of the kernel sources this library was built from: PMC figures measured on another version of the
    kernels are not quoted as if they belonged to this one."""
    import hashlib

    h = hashlib.sha256()
    # what the device code is compiled from (field, group laws, kernels, grouping, launchers, every per-curve unit); host-only files --
    # the engine, the fold, the staging pipeline, test scaffolding -- can change without invalidating a counter measurement
    kernel_files = ("curve.hpp", "digits.hpp", "field_consts.inc", "fp28.hpp", "fp2pair.hpp", "laws.hpp", "launch.hpp", "launch_impl.hpp",
                    "launch_pair_impl.hpp", "msm_kernels.hpp", "msm_types.hpp", "partition.hpp", "partition_plan.hpp", "te.hpp",
                    "kernels_377g1.hip", "kernels_377g2.hip", "kernels_377g2p.hip", "kernels_377te.hip", "kernels_381g1.hip", "kernels_381g2.hip",
                    "kernels_381g2p.hip", "partition.hip")
    for name in kernel_files:
        f = os.path.join(ROOT, "2022-entries_amd", "csrc", name)
        h.update(name.encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# ---- the run's own clock and power ----------------------------------------------------------------------------------------------
class Telemetry:
    """Shader clock (MHz) and socket power (W) of one GPU, sampled by a thread while a timed loop runs.  Backends, first that works:
    the amdgpu hwmon files in sysfs (freq1_input / power1_average|power1_input; pp_dpm_sclk's starred level when there is no
    freq1_input), then the amdsmi Python package.  None available: every figure is None and `source` says why."""

    def __init__(self, torch_device_index, hz=20.0):
        self.period = 1.0 / hz
        self.source = None
        self.why_not = []
        self._read = None
        self._thread = None
        self._stop = threading.Event()
        self.samples = []
        try:
            self._init_sysfs(torch_device_index)
        except Exception as e:   # never lose a bench line to telemetry
            self.why_not.append("sysfs: %r" % (e,))
        if self._read is None:
            try:
                self._init_amdsmi(torch_device_index)
            except Exception as e:
                self.why_not.append("amdsmi: %r" % (e,))

    @staticmethod
    def _bdf(idx):
        import torch

        p = torch.cuda.get_device_properties(idx)
        if all(hasattr(p, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        return None

    def _init_sysfs(self, idx):
        cards = []
        for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
            if hw and (os.path.exists(os.path.join(dev, "pp_dpm_sclk")) or os.path.exists(os.path.join(hw[0], "freq1_input"))):
                cards.append((os.path.basename(os.path.realpath(dev)), dev, hw[0]))
        if not cards:
            self.why_not.append("sysfs: no amdgpu card with hwmon under /sys/class/drm")
            return
        bdf = self._bdf(idx)
        pick = [c for c in cards if bdf and c[0] == bdf]
        if not pick:
            if len(cards) != 1:
                self.why_not.append("sysfs: %d cards, none matches the device's PCI address %s" % (len(cards), bdf))
                return
            pick = cards
        _, dev, hw = pick[0]
        f_clk = os.path.join(hw, "freq1_input")
        f_dpm = os.path.join(dev, "pp_dpm_sclk")
        f_pow = next((p for p in (os.path.join(hw, "power1_average"), os.path.join(hw, "power1_input")) if os.path.exists(p)), None)

        def read():
            mhz = watts = None
            try:
                if os.path.exists(f_clk):
                    mhz = int(open(f_clk).read()) / 1e6
                else:
                    for ln in open(f_dpm):
                        if "*" in ln:
                            mhz = float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            except (OSError, ValueError, IndexError):
                pass
            try:
                if f_pow:
                    watts = int(open(f_pow).read()) / 1e6
            except (OSError, ValueError):
                pass
            return mhz, watts

        if read() == (None, None):
            self.why_not.append("sysfs: %s has neither a readable clock nor power file" % hw)
            return
        self._read = read
        self.source = "sysfs " + hw + (" (freq1_input" if os.path.exists(f_clk) else " (pp_dpm_sclk") + (", %s)" % os.path.basename(f_pow) if f_pow else ")")

    def _init_amdsmi(self, idx):
        import amdsmi   # noqa: F401  (optional)

        amdsmi.amdsmi_init()
        handles = amdsmi.amdsmi_get_processor_handles()
        bdf = self._bdf(idx)
        h = None
        for cand in handles:
            try:
                if bdf and amdsmi.amdsmi_get_gpu_device_bdf(cand).lower() == bdf:
                    h = cand
            except Exception:
                pass
        if h is None:
            if len(handles) != 1 and idx >= len(handles):
                self.why_not.append("amdsmi: no handle for device %d" % idx)
                return
            h = handles[idx if idx < len(handles) else 0]

        def read():
            mhz = watts = None
            try:
                ci = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                mhz = float(ci.get("clk", ci.get("cur_clk")))
            except Exception:
                pass
            try:
                pi = amdsmi.amdsmi_get_power_info(h)
                for k in ("current_socket_power", "average_socket_power", "socket_power"):
                    if isinstance(pi.get(k), (int, float)) and pi[k] > 0:
                        watts = float(pi[k])
                        break
            except Exception:
                pass
            return mhz, watts

        if read() == (None, None):
            self.why_not.append("amdsmi: neither clock nor power readable")
            return
        self._read = read
        self.source = "amdsmi"

    def start(self):
        self.samples = []
        if self._read is None:
            return
        self._stop.clear()

        def loop():
            while not self._stop.is_set():
                self.samples.append(self._read())
                self._stop.wait(self.period)

        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        """-> {"clock_MHz_mean", "clock_MHz_min", "clock_MHz_max", "power_W_mean", "power_W_max", "samples"} (None where unknown)"""
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None
        clk = [c for c, _ in self.samples if c]
        pw = [p for _, p in self.samples if p]
        return {"clock_MHz_mean": sum(clk) / len(clk) if clk else None, "clock_MHz_min": min(clk) if clk else None,
                "clock_MHz_max": max(clk) if clk else None, "power_W_mean": sum(pw) / len(pw) if pw else None,
                "power_W_max": max(pw) if pw else None, "samples": len(self.samples)}

    def describe(self):
        return self.source or ("unavailable: " + "; ".join(self.why_not))


def hbm_probe(torch, device, gib=4):
    """confirm the HBM figure on the box.  One device-to-device copy of `gib` GiB (read + write = 2 x gib GiB of traffic,
    hipMemcpyDtoD under torch's copy_) and one read-only sweep of the same buffer, each the best of 5, HIP events on the current stream."""
    try:
        nbytes = gib << 30
        src = torch.empty(nbytes // 8, dtype=torch.int64, device=device)
        src.fill_(0x0101010101010101)
        dst = torch.empty_like(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best_copy = best_read = 0.0
        dst.copy_(src)
        for _ in range(5):
            e0.record()
            dst.copy_(src)
            e1.record()
            e1.synchronize()
            best_copy = max(best_copy, 2 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        src.sum()
        for _ in range(5):
            e0.record()
            src.sum()
            e1.record()
            e1.synchronize()
            best_read = max(best_read, nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del src, dst
        torch.cuda.empty_cache()
        return {"copy_GBps": best_copy, "read_GBps": best_read, "what": f"{gib} GiB device-to-device copy (bytes read + written) and a read-only sweep, best of 5"}
    except Exception as e:
        return {"error": repr(e)}


def stage_roofline(n, tm, ms, cid, geom, pmc_stages=None):
    """The HBM-bound stages against HBM: STRUCTURAL bytes each stage has to move once, over its measured time,
    as a fraction of the achievable streaming rate.  digits = level 1 of the grouping (scalars read twice, the tile x bin count
    matrix written once and read/written by the column scan and read by the scatter, the entries written); sort = the generic
    passes (entries in from HBM, entries out; the fused kernel's second read comes from cache and is not counted); bucket_reduce =
    every bucket read once.  The geometry -- level-1 bins, passes, bucket sets -- is the ENGINE's (mi355_msm_query after the run), not
    a re-derivation of its heuristics."""
    c, E = tm["window_bits"], tm["entries"]
    ntiles = -(-n // 8192)
    matrix = ntiles * geom["l1_bins"] * 4
    xyzz = 448 if cid >= 2 else 224
    stages = {"digits": 2 * 32 * n + 5 * matrix + 8 * E, "sort": geom["group_passes"] * 16 * E,
              "bucket_reduce": geom["bucket_windows"] * (1 << (c - 1)) * xyzz}
    out = {}
    for k, b in stages.items():
        if ms.get(k):
            gbps = b / (ms[k] * 1e-3) / 1e9
            out[k] = {"structural_bytes": b, "ms": ms[k], "GBps": gbps, "frac_of_achievable": gbps / HBM_ACHIEVABLE_GBPS}
    out["achievable_GBps"] = HBM_ACHIEVABLE_GBPS
    out["geometry"] = geom
    if pmc_stages:
        # the same stages from COUNTERS (profiles/<round>_pmc_k_accumulate*.json "stages": FETCH_SIZE / WRITE_SIZE per kernel of one bench step,
        # measured on these kernel sources under this plan): what the kernels asked of the fabric, against the structural bytes above
        by_stage = {}
        for kern, row in pmc_stages.items():
            d = by_stage.setdefault(row["stage"], {"hbm_bytes_estimate": 0.0, "kernel_ms": 0.0, "kernels": []})
            d["hbm_bytes_estimate"] += row.get("hbm_bytes_estimate") or 0.0
            d["kernel_ms"] += row.get("ms_per_step") or 0.0
            d["kernels"].append(kern)
            if row.get("valu_busy_fraction") is not None and kern in ("k_bucket_reduce", "k_pass_scatter", "k_l1_scatter"):
                d["valu_busy_fraction"] = row["valu_busy_fraction"]
        for k, d in by_stage.items():
            if k in out and d["kernel_ms"]:
                d["GBps"] = d["hbm_bytes_estimate"] / (d["kernel_ms"] * 1e-3) / 1e9
                d["counter_over_structural_bytes"] = d["hbm_bytes_estimate"] / out[k]["structural_bytes"]
                out[k]["counters"] = d
    return out


def timed(fn, reps):
    """Wall ms per call of fn() (each call returns with the result on the host, i.e. it is synchronous)."""
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps * 1e3, r


def cpu_baseline(cid, bases_np, scalars_np, sample, threads):
    """Time the oracle (arkworks-algorithm restatement) on `sample` pairs; returns (pairs/s, result bytes, seconds)."""
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    out = ctypes.create_string_buffer(288 if cid >= 2 else 144)
    t0 = time.perf_counter()
    rc = lib.oracle_msm(cid, bases_np.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(bases_np.shape[1]),
                        scalars_np.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(sample), out, threads)
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError("oracle_msm failed")
    return sample / dt, out.raw, dt


def ark_ec_probe(npow, budget_s=900):
    """"if cargo is present on the GPU box, additionally run real ark-ec via the committed Rust shim and report both".
    Runs rust/benches/msm.rs (`cargo bench --offline`: no network on the box) and parses ARK_EC_CPU_MS.  This image has no cargo, so
    on it the probe reports exactly that."""
    cargo = shutil.which("cargo")
    if not cargo:
        return {"available": False, "probe": "no `cargo` on PATH (this image ships no Rust toolchain): cpu_baseline.kind stays \"port\""}
    env = dict(os.environ, BENCH_NPOW=str(npow), BENCH_REPS="1", MI355_MSM_LIB_DIR=os.path.join(ROOT, "2022-entries_amd"),
               LD_LIBRARY_PATH=os.path.join(ROOT, "2022-entries_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    try:
        r = subprocess.run([cargo, "bench", "--offline", "--manifest-path", os.path.join(ROOT, "rust", "Cargo.toml")], env=env, capture_output=True,
                           text=True, timeout=budget_s)
    except subprocess.TimeoutExpired:
        return {"available": True, "error": f"cargo bench did not finish in {budget_s} s"}
    kv = {}
    for ln in r.stdout.splitlines():
        for tok in ln.split():
            if "=" in tok:
                k, v = tok.split("=", 1)
                kv[k] = v
    if r.returncode != 0 or "ARK_EC_CPU_MS" not in kv:
        return {"available": True, "error": "cargo bench failed (offline build without a vendored registry?)", "rc": r.returncode, "stderr_tail": r.stderr[-600:]}
    ms = float(kv["ARK_EC_CPU_MS"])
    return {"available": True, "kind": "ark-ec", "value": (1 << npow) / ms * 1e3, "unit": "pairs/s", "ms": ms, "cores": int(kv.get("HOST_THREADS", 0)),
            "sample": f"ark-ec 0.3 VariableBaseMSM::multi_scalar_mul, 2^{npow} pairs, rayon over the host's threads (rust/benches/msm.rs)",
            "gpu_equals_ark_ec": kv.get("GPU_EQUALS_ARK_EC") == "1",
            "harness_four_batches_ms": float(kv["MI355_MSM_MS_PER_4_BATCHES"]) if "MI355_MSM_MS_PER_4_BATCHES" in kv else None}


class Env:
    """What the measurement functions share: modules, device, the synthetic workload."""


# ---- secondary measurements of the N = 1 line: each owns the contexts it creates -------------------------------------------------
def measure_host_scalars(E, ctx, result):
    """SURVEY 8(d)'s PRIMARY metric is what the reference bench times: bases resident, scalars in HOST memory, 4 batches
    (P1A combined-top-solutions/benches/msm.rs:21,27-35).  Borrows the headline context (its bases are the workload's)."""
    torch = E.torch
    sc_np = E.scalars.cpu().numpy()                                   # pageable host memory
    sc_pin = torch.from_numpy(sc_np).pin_memory()
    ms1_page, r1 = timed(lambda: ctx.run(sc_np)[0], 3)
    ms1_pin, r1p = timed(lambda: ctx.run(sc_pin)[0], 3)
    out = {"one_batch_ms": {"pageable": ms1_page, "pinned": ms1_pin}, "same_result_as_device_scalars": r1 == result and r1p == result}
    sc4 = torch.cat([uniform_scalars(E.n, r_top(E.cid), E.device, seed=4000 + b) for b in range(4)])
    sc4_np = sc4.cpu().numpy()
    sc4_pin = torch.from_numpy(sc4_np).pin_memory()
    ms4_dev, r4 = timed(lambda: ctx.run(sc4), 2)
    ms4_page, r4p = timed(lambda: ctx.run(sc4_np), 2)
    ms4_pin, r4q = timed(lambda: ctx.run(sc4_pin), 2)
    out["four_batches_ms"] = {"pageable": ms4_page, "pinned": ms4_pin, "device_resident": ms4_dev,
                              "what": "the ZPrize workload: 4 x 2^%d scalars over one base vector, one call" % E.npow, "same_results": r4 == r4p == r4q}
    return out


def measure_assume_subgroup(E, ctx, result):
    """The winners' top-bit trick as a context option (off by default: it needs every base in the order-r subgroup, which this
    generator's bases are): scalars above r/2 run as (r - k)(-P), CMB ProcessSignedDigits.cu:123-128.  Borrows the headline context and
    restores the option."""
    ctx.set_option("assume_subgroup", 1)
    try:
        ms_f, rf = timed(lambda: ctx.run(E.scalars)[0], E.args.steps)
        tf = ctx.last_timings()
    finally:
        ctx.set_option("assume_subgroup", 0)
    return {"ms_per_step": ms_f, "window_bits": tf["window_bits"], "accumulate_ms": tf["accumulate"], "same_result": rf == result,
            "what": "same workload with the context option assume_subgroup = 1 (not the headline: the default path is exact for any curve point)"}


def measure_xyzz(E, result):
    """The group law north_star names (XYZZ, 8M + 2S) on the same BLS12-377 workload."""
    cx = E.ea.MultiScalarMultContext(E.args.curve, device=E.local_rank)
    try:
        cx.set_option("twisted_edwards", 0)
        cx.set_bases(E.tile.repeat(E.n // E.distinct, 1).contiguous())
        ms_x, rx = timed(lambda: cx.run(E.scalars)[0], E.args.steps)
        return {"xyzz_ms_per_step": ms_x, "xyzz_accumulate_ms": cx.last_timings()["accumulate"], "xyzz_same_result": rx == result}
    finally:
        cx.close()


def measure_stateless(E, result):
    """One stateless call: host bases -> upload -> conversion -> MSM -> teardown (a pipeline since round 3: slices cross PCIe through a
    pinned ring while earlier ones compute, csrc/msm_stateless.hpp).  The library's own pooled context; nothing to close here."""
    np, ea = E.np, E.ea
    sc_np = E.scalars.cpu().numpy()
    bases_host = np.ascontiguousarray(np.tile(E.base_tile, (E.n // E.distinct, 1)))
    out = {}
    t_s = time.perf_counter()
    rs = ea.msm(bases_host, sc_np, E.args.curve)
    out["stateless_first_ms"] = (time.perf_counter() - t_s) * 1e3     # first call of the process: allocates the pinned ring
    first_stats = ea.last_stateless()
    warm = []
    for _ in range(3):
        t_s = time.perf_counter()
        rs2 = ea.msm(bases_host, sc_np, E.args.curve)
        warm.append((time.perf_counter() - t_s) * 1e3)
    warm.sort()
    out["stateless_ms"] = warm[1]
    out["stateless_ms_all"] = warm
    # operands the process has never touched through HIP before (fresh pages): what a cold caller sees
    bases_cold = bases_host.copy()
    sc_cold = sc_np.copy()
    t_s = time.perf_counter()
    rs3 = ea.msm(bases_cold, sc_cold, E.args.curve)
    out["stateless_fresh_operands_ms"] = (time.perf_counter() - t_s) * 1e3
    out["stateless_pipeline"] = {"first_call": first_stats, "fresh_operands": ea.last_stateless()}
    out["stateless_same_result"] = rs == result and rs2 == result and rs3 == result
    out["stateless_what"] = ("mi355_msm(): %.1f GB of bases and %.1f GB of scalars from pageable host memory, everything included "
                             "(upload, conversion, MSM on the XYZZ law, teardown)") % (bases_host.nbytes / 1e9, sc_np.nbytes / 1e9)
    out["stateless_pcie_floor_ms"] = (bases_host.nbytes + sc_np.nbytes) / 57e9 * 1e3
    return out


def measure_secondary_configs(E):
    """BASELINE.json configs[2] and configs[4] (+ the BLS12-381 twin of the latter), so that the driver's line carries them: the second
    384-bit prime (no Edwards form: XYZZ) and G2 over Fq2.  Scalars resident in HBM, as for `value`."""
    torch, ea = E.torch, E.ea
    sec = {}
    for name, cname, npow2 in (("bls12_381_g1_2^26", "bls12_381_g1", 26), ("bls12_377_g2_2^24", "bls12_377_g2", 24), ("bls12_381_g2_2^24", "bls12_381_g2", 24)):
        c2 = None
        try:
            cid2, n2 = ea.CURVE_IDS[cname], 1 << npow2
            tile2 = torch.from_numpy(ea.generate_points(E.distinct, distinct=E.distinct, seed=0x5A5052495A45 + cid2, curve=cname)).to(E.device)
            sc2 = uniform_scalars(n2, r_top(cid2), E.device, seed=99 + cid2)
            c2 = ea.MultiScalarMultContext(cname, device=E.local_rank)
            c2.set_bases(tile2.repeat(n2 // E.distinct, 1).contiguous())
            ms2, _ = timed(lambda: c2.run(sc2)[0], 3)
            tm2 = c2.last_timings()
            k_ms = tm2["accumulate"] / max(tm2["launches"], 1)
            sec[name] = {"ms_per_step": ms2, "value": n2 / ms2 * 1e3, "unit": "pairs/s", "window_bits": tm2["window_bits"],
                         "stage_ms": {k: tm2[k] for k in ("digits", "sort", "accumulate", "segreduce", "bucket_reduce", "host_fold")},
                         "roofline_frac_hbm": BYTES_PER_PAIR[cid2] * n2 / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            g2p = c2.query("g2_paired") if cid2 >= 2 else 0
            if cid2 >= 2:
                sec[name]["g2_paired"] = g2p   # bit mask of the kernels that run two lanes per point (csrc/fp2pair.hpp)
            # the integer roofline of this config's accumulation (as `roofline.integer` of the headline): multiply-adds per mixed addition are
            # pinned on the ISA (tests/test_isa.py, tools/isa_histogram.py)
            mads2 = {1: 3542, 2: 11480 if (g2p & 1) else 11088, 3: 11760 if (g2p & 1) else 11368}[cid2]
            rate2 = mads2 * tm2["entries"] / (k_ms * 1e-3)
            sec[name]["integer"] = {"mads_per_mixed_add": mads2, "lane_mads_per_s": rate2, "peak_lane_mads_per_s": MAD_PEAK, "frac": rate2 / MAD_PEAK}
            if cname in ("bls12_377_g2", "bls12_381_g1"):
                # BASELINE configs[4] once more with "precompute" = 2 (auto: tables from the free HBM, init untimed): non-default, reported beside
                c2.close()
                c2 = ea.MultiScalarMultContext(cname, device=E.local_rank)
                c2.set_option("precompute", 2)
                t_i = time.perf_counter()
                c2.set_bases(tile2.repeat(n2 // E.distinct, 1).contiguous())
                torch.cuda.synchronize()
                t_i = time.perf_counter() - t_i
                ms3, _ = timed(lambda: c2.run(sc2)[0], 3)
                tm3 = c2.last_timings()
                sec[name]["with_precompute_auto"] = {"ms_per_step": ms3, "table_levels": c2.query("table_levels"), "window_bits": tm3["window_bits"],
                                                     "table_bytes": c2.query("base_bytes"), "init_s": t_i,
                                                     "stage_ms": {k: tm3[k] for k in ("digits", "sort", "accumulate", "segreduce", "bucket_reduce")}}
            del tile2, sc2
        except Exception as e:
            sec[name] = {"error": repr(e)}
        finally:
            if c2 is not None:
                c2.close()
    return sec


def measure_small_latency(E):
    """Latency of small inputs (BASELINE.json configs[0] is 2^16): wall ms of one MSM, median of 15, bases and
    scalars resident, with the device-side share."""
    torch, ea = E.torch, E.ea
    lat = {}
    for npow2 in (10, 16, 20):
        c2 = None
        try:
            n2 = 1 << npow2
            c2 = ea.MultiScalarMultContext(E.args.curve, device=E.local_rank)
            c2.set_bases(E.tile[:n2].contiguous() if n2 <= E.distinct else E.tile.repeat(n2 // E.distinct, 1).contiguous())
            sc2 = uniform_scalars(n2, r_top(E.cid), E.device, seed=7)
            for _ in range(3):
                c2.run(sc2)
            ts = []
            for _ in range(15):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                c2.run(sc2)
                ts.append(time.perf_counter() - t1)
            ts.sort()
            tm2 = c2.last_timings()
            lat["2^%d" % npow2] = {"wall_ms": ts[7] * 1e3, "device_ms": tm2["total"], "host_fold_ms": tm2["host_fold"], "window_bits": tm2["window_bits"]}
        except Exception as e:
            lat["2^%d" % npow2] = {"error": repr(e)}
        finally:
            if c2 is not None:
                c2.close()
    return lat


def measure_cpu_baseline(E, result):
    np, ea, torch = E.np, E.ea, E.torch
    sample = min(E.n, 1 << E.args.cpu_sample_pow)
    cores = os.cpu_count() or 1
    if E.cid >= 2:
        sample = min(sample, 1 << 21)   # Fp2 arithmetic is ~3x slower on the CPU too
    bases_np = np.ascontiguousarray(np.tile(E.base_tile, (max(1, sample // E.distinct), 1))[:sample])
    scal_np = E.scalars[:sample].cpu().numpy()
    c = 3 if sample < 32 else (((sample - 1).bit_length()) * 69 // 100 + 2)
    windows = -(-(255 if E.cid in (1, 3) else 253) // c)
    threads = min(windows, cores)
    v, cpu_res, dt = cpu_baseline(E.cid, bases_np, scal_np, sample, threads)
    # same sample on the GPU: a parity spot-check next to the number (`result` is the headline run's point when the
    # sample is the whole workload; otherwise a context over the sample's bases recomputes it)
    if sample == E.n:
        gpu_res = result
    else:
        cs = ea.MultiScalarMultContext(E.args.curve, device=E.local_rank)
        try:
            cs.set_bases(torch.from_numpy(bases_np).to(E.device))
            gpu_res = cs.run(E.scalars[:sample].contiguous())[0]
        finally:
            cs.close()
    out = {"value": v, "unit": "pairs/s", "cores": threads, "kind": "port",
           "sample": f"first 2^{sample.bit_length() - 1} pairs of the same workload, {dt:.1f} s, "
                     f"arkworks-algorithm restatement (c={c}, one thread per window), host has {cores} cores",
           "gpu_matches_cpu_on_sample": gpu_res == cpu_res}
    if E.cid == 0:
        out["ark_ec"] = ark_ec_probe(min(E.npow, E.args.cpu_sample_pow))
    return out


def measure_tables(E, result, with_four_batches):
    """The reference's own convention (P1A combined-top-solutions/benches/msm.rs:21,27-35; CMB MSM.cu:380-383): `init` builds the
    precomputed tables untimed, then FOUR batches of scalars come from pageable host memory.  Reported next to the headline, never as
    it.  table_levels = k: k levels 2^(c G j) P, windows g, g + G, ... share bucket set g (yrrid: k = 6, G = 2); 0 = a level per window
    (one bucket set); "auto" = option precompute = 2: the context picks k from the HBM that is free at set_bases.  Per row: ms per MSM
    with resident scalars, HBM held by the tables, init seconds, and the ZPrize workload itself (4 x 2^npow scalars from pageable host
    memory, one call)."""
    torch, ea = E.torch, E.ea
    sc4p = None
    if with_four_batches:
        sc4p = torch.cat([uniform_scalars(E.n, r_top(E.cid), E.device, seed=4000 + b) for b in range(4)]).cpu().numpy()
    table = {}
    seen_levels = {}
    for levels in (("auto", 0, 6, 3, 2) if E.npow >= 20 else ("auto", 0, 6)):
        name = "all" if levels == 0 else str(levels)
        ctx2 = None
        try:
            ctx2 = ea.MultiScalarMultContext(E.args.curve, device=E.local_rank)
            if levels == "auto":
                ctx2.set_option("precompute", 2)
            else:
                ctx2.set_option("precompute", 1)
                ctx2.set_option("table_levels", levels)
            t_i = time.perf_counter()
            ctx2.set_bases(E.tile.repeat(E.n // E.distinct, 1).contiguous())
            torch.cuda.synchronize()
            t_i = time.perf_counter() - t_i
            built = ctx2.query("table_levels")
            if levels == "auto" and built <= 1:
                table[name] = {"table_levels": built, "what": "precompute = auto chose NO tables for this size / this much free HBM: the context is the headline's"}
                continue
            if levels != "auto" and built in seen_levels.values() and seen_levels.get("auto") == built:
                # the automatic choice already measured this very shape
                table[name] = dict(table["auto"], same_shape_as="auto")
                continue
            seen_levels[name] = built
            r2 = ctx2.run(E.scalars)[0]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(E.args.steps):
                r2 = ctx2.run(E.scalars)[0]
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            tm2 = ctx2.last_timings()
            row = {"ms_per_step": dt2 / E.args.steps * 1e3, "value": E.n * E.args.steps / dt2, "unit": "pairs/s", "init_s": t_i,
                   "table_levels": built, "window_bits": tm2["window_bits"], "windows": tm2["windows"],
                   "bucket_sets": ctx2.query("bucket_windows"),
                   "table_bytes": ctx2.query("base_bytes"), "group_law": "twisted Edwards" if ctx2.query("twisted_edwards") else "XYZZ",
                   "stage_ms": {k: tm2[k] for k in ("digits", "sort", "accumulate", "segreduce", "bucket_reduce")},
                   "same_result_as_headline_path": r2 == result}
            if sc4p is not None:
                ms4, _ = timed(lambda: ctx2.run(sc4p), 2)
                row["four_batches_from_pageable_host_ms"] = ms4
            table[name] = row
        except Exception as e:  # e.g. not enough free HBM for the tables
            table[name] = {"error": str(e)}
        finally:
            if ctx2 is not None:
                ctx2.close()
    del sc4p
    return dict(table.get("all", {}), by_table_levels=table, auto=table.get("auto"),
                what="context option precompute = 1 (init untimed, as the reference's bench does); by_table_levels: k -> the same workload with k "
                     "table levels, \"auto\" -> precompute = 2 (the context's own choice from free HBM); four_batches_from_pageable_host_ms is the "
                     "ZPrize workload (reference: 2200-2300 ms on an A40 with 6 levels)")


def quoted_pmc(E, tm, te_path, total_npow):
    """HBM traffic and VALU occupancy of the dominant kernel come from separate rocprofv3 --pmc passes (committed under profiles/);
    counters need their own pass, so this run does NOT measure them: `traffic_from` says where the figure was measured, and it is only
    quoted when that pass ran on the very kernel sources this library was built from, under the very plan this run executes."""
    args = E.args
    traffic = traffic_raw = traffic_from = valu = stages = None
    pmc_rel = os.path.join("profiles", "%s_pmc_k_accumulate%s.json" % (PMC_ROUND, {0: "", 1: "_381", 2: "_g2", 3: "_381g2"}[E.cid]))
    pmc_path = os.path.join(ROOT, pmc_rel)
    if (args.npow == (24 if E.cid >= 2 else 26) and not total_npow and not args.window_bits and not args.lane_entries and not args.precompute
            and os.path.exists(pmc_path)):
        pmc = json.load(open(pmc_path))
        sha = kernel_source_sha16()
        plan_now = {"window_bits": tm["window_bits"], "windows": tm["windows"], "lane_entries": tm["lane_entries"], "lanes": tm["lanes"],
                    "group_law": "extended twisted Edwards (7M mixed add)" if te_path else "XYZZ (8M+2S mixed add)"}
        plan_then = dict(pmc.get("plan") or {})
        if E.cid >= 2 and plan_then.get("lanes") == 2 * plan_now["lanes"]:
            plan_then["lanes"] = plan_now["lanes"]       # (rocprof counts hardware lanes: two per walking lane in the paired G2 kernels)
        if plan_then.get("lanes") == -(-plan_now["lanes"] // 256) * 256:
            plan_then["lanes"] = plan_now["lanes"]       # (rocprof reports the grid: whole blocks of 256 threads)
        if pmc.get("kernel_source_sha16") == sha and plan_then != plan_now:
            traffic_from = f"not quoted: {pmc_rel} was measured under the plan {pmc.get('plan')}, this run uses {plan_now}"
        elif pmc.get("kernel_source_sha16") == sha:
            # corrected = FETCH_SIZE / WRITE_SIZE rescaled by the ratios tools/calib_fetch.hip measured on the kernel's own access
            # shapes (gfx950 tallies 64 B per fabric request, whether it is a 64- or a 128-byte one)
            traffic = pmc.get("traffic_bytes_corrected") or pmc["traffic_bytes_raw"]
            traffic_raw = pmc["traffic_bytes_raw"]
            traffic_from = (f"{pmc_rel}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this workload, same kernel sources ({sha}); "
                            f"corrected with {pmc.get('traffic_model', {}).get('calibration', {}).get('file')}")
            valu = {"busy_fraction": pmc["derived"]["valu_busy_fraction"], "effective_clock_GHz": pmc["derived"]["effective_clock_GHz"],
                    "valu_instr_per_mixed_add": pmc["derived"]["valu_instr_per_mixed_add"], "from": pmc_rel,
                    "note": "counters of the committed profile's box; THIS run's clock is config.clock_MHz_*"}
            stages = pmc.get("stages")
        else:
            traffic_from = f"not quoted: {pmc_rel} was measured on kernel sources {pmc.get('kernel_source_sha16')}, this library is built from {sha}"
    return traffic, traffic_raw, traffic_from, valu, stages


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--npow", type=int, default=26, help="log2 pairs per GPU (26 = ZPrize prize1-msm canonical size)")
    ap.add_argument("--total-npow", type=int, default=-1,
                    help="log2 pairs of the WHOLE job, split evenly over the GPUs (overrides --npow); default: 28 when --gpus 8 "
                         "(BASELINE.json configs[3]), otherwise unset (2^npow per GPU); 0 = never")
    ap.add_argument("--curve", default="bls12_377_g1", choices=["bls12_377_g1", "bls12_381_g1", "bls12_377_g2", "bls12_381_g2"])
    ap.add_argument("--only", default="all", choices=["all", "headline"],
                    help="headline = the timed loop, its re-takes, telemetry and the HBM probe only (seconds): the number the driver times, "
                         "re-taken several times per lease")
    ap.add_argument("--repeat", type=int, default=-1,
                    help="re-take the K-step timed loop this many more times (`headline_samples`); default 2 at N = 1 (4 with --only headline), 0 otherwise")
    ap.add_argument("--cpu-sample-pow", type=int, default=26,
                    help="log2 pairs of the CPU-baseline sample (0 = skip); 26 = the whole workload once, about a minute of host time")
    ap.add_argument("--extras", type=int, default=1,
                    help="at N = 1 also measure, in the same run, SURVEY 8(d)'s primary metric and its neighbours: scalars in HOST memory "
                         "(1 and 4 batches, pageable and pinned), the XYZZ group law on BLS12-377, one stateless mi355_msm() call")
    ap.add_argument("--logical-shards", type=int, default=0,
                    help="single-process --gpus N on a box with fewer GPUs: place the N shards on the visible devices round-robin")
    ap.add_argument("--precompute", type=int, default=0, help="1 = context with precomputed 2^(c w) P tables (row f1; init untimed); 2 = auto")
    ap.add_argument("--also-precompute", type=int, default=1,
                    help="at N = 1 also time contexts with precomputed tables (reported as a secondary object, never as `value`)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for --gpus > 1 (nccl = RCCL over xGMI; gloo only to rehearse the multi-rank path on one GPU)")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--lane-entries", type=int, default=0)
    ap.add_argument("--assume-subgroup", type=int, default=0,
                    help="1 = context option assume_subgroup (scalars above r/2 run as (r - k)(-P): the winners' top-bit trick, valid for bases in the r-torsion)")
    ap.add_argument("--hbm-probe", type=int, default=1, help="0 = skip the device-to-device copy probe (roofline.peak_measured)")
    args = ap.parse_args()
    headline_only = args.only == "headline"
    if headline_only:
        args.extras = args.also_precompute = args.cpu_sample_pow = 0

    import numpy as np
    import torch
    import torch.distributed as dist

    import entries_amd as ea

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # --gpus N without a launcher (WORLD_SIZE unset): ONE process drives N GPUs through the sharded context of the C ABI
    # (mi355_msm_create_sharded: per-device host threads, RCCL all-gather of the partials, host fold).  Under
    # torch.distributed.run the same N GPUs are one process each (dist.py), which is what the driver launches.
    c_sharded = world == 1 and args.gpus > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MSM path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()      # rehearsal: several ranks may share one GPU
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")
    coll_device = device if args.backend == "nccl" else None
    single = world == 1 and not c_sharded
    repeat = args.repeat if args.repeat >= 0 else ((4 if headline_only else 2) if single else 0)

    cid = ea.CURVE_IDS[args.curve]
    n_ranks = args.gpus if c_sharded else world
    total_npow = args.total_npow
    if total_npow < 0:
        # N = 8 defaults to BASELINE.json configs[3]: FOUR times the single-GPU problem sharded eight ways (2^28 = 8 x 2^25 at the default
        # --npow 26).  A smaller --npow keeps the shape (2^(npow+2) over 8 GPUs) so the same code path can be rehearsed at small sizes.
        total_npow = args.npow + 2 if (n_ranks == 8 and args.curve == "bls12_377_g1") else 0
    weak_secondary = None
    if total_npow:
        if n_ranks & (n_ranks - 1) or (1 << total_npow) < n_ranks:
            raise SystemExit("--total-npow needs a power-of-two number of GPUs")
        weak_secondary = args.npow if args.total_npow < 0 else None      # the default N = 8 line also carries the weak-scaling point
        args.npow = total_npow - (n_ranks.bit_length() - 1)
    n = 1 << args.npow
    distinct = min(n, 1 << 15)

    # the box's own HBM rate, before the workload takes the memory (rank 0 of a single-GPU run: the probe is about the chip, not the job)
    probe = hbm_probe(torch, device) if (single and args.hbm_probe) else None
    telemetry = Telemetry(local_rank)

    # synthetic inputs in the reference generator's shape: 2^15 distinct subgroup points replicated to n,
    # uniform scalars below r; every rank has its own slice of the global problem (different scalars per rank)
    base_tile = ea.generate_points(distinct, distinct=distinct, seed=0x5A5052495A45 + cid, curve=args.curve)
    tile = torch.from_numpy(base_tile).to(device)
    bases = tile.repeat(n // distinct, 1).contiguous()
    scalars = uniform_scalars(n, r_top(cid), device, seed=1234 + rank)
    devs = None
    if c_sharded:
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and not args.logical_shards:
            raise SystemExit(f"--gpus {args.gpus} but {ndev} device(s) visible (use --logical-shards 1 to rehearse on fewer)")
        devs = [g % ndev for g in range(args.gpus)]
        ctx = ea.MultiScalarMultContext(args.curve, devices=devs)
        # weak scaling: 2^npow pairs PER GPU; the global problem is args.gpus times as large
        bases = bases.repeat(args.gpus, 1)
        scalars = torch.cat([uniform_scalars(n, r_top(cid), device, seed=1234 + g) for g in range(args.gpus)])
    else:
        ctx = ea.MultiScalarMultContext(args.curve, device=local_rank)
    if args.window_bits:
        ctx.set_option("window_bits", args.window_bits)
    if args.precompute:
        ctx.set_option("precompute", args.precompute)
    if args.assume_subgroup:
        ctx.set_option("assume_subgroup", 1)
    t_init = time.perf_counter()
    ctx.set_bases(bases)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t_init
    del bases
    if args.lane_entries:
        ctx.set_option("lane_entries", args.lane_entries)

    E = Env()
    E.args, E.np, E.torch, E.ea = args, np, torch, ea
    E.device, E.local_rank, E.cid, E.n, E.npow, E.distinct = device, local_rank, cid, n, args.npow, distinct
    E.tile, E.base_tile, E.scalars = tile, base_tile, scalars

    def step():
        partial = ctx.run(scalars)[0]
        if world > 1:
            return ea.fold_partials(ea.all_gather_partials(partial, device=coll_device), args.curve)
        return partial

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop():
        """EXACTLY --steps steps between two fences; -> (seconds, result, summed stage ms, accumulate ms, launches, last timings, telemetry)"""
        stage = {}
        acc = launches = 0
        fence()
        telemetry.start()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
            tmx = ctx.last_timings()
            acc += tmx["accumulate"]
            launches += tmx["launches"]
            for k in STAGES:
                stage[k] = stage.get(k, 0.0) + tmx[k]
        fence()
        dt = time.perf_counter() - t0
        return dt, res, stage, acc, launches, tmx, telemetry.stop()

    for _ in range(args.warmup):
        step()
    elapsed, result, stage_ms, acc_ms, acc_launches, tm, tele = timed_loop()
    # the same loop again, `repeat` times: what one sample of a power-limited kernel is worth
    samples = [{"ms_per_step": elapsed / args.steps * 1e3, "accumulate_ms": acc_ms / max(acc_launches, 1), "clock_MHz_mean": tele["clock_MHz_mean"],
                "power_W_mean": tele["power_W_mean"]}]
    for _ in range(repeat):
        dt_r, res_r, _, acc_r, launches_r, _, tele_r = timed_loop()
        samples.append({"ms_per_step": dt_r / args.steps * 1e3, "accumulate_ms": acc_r / max(launches_r, 1), "clock_MHz_mean": tele_r["clock_MHz_mean"],
                        "power_W_mean": tele_r["power_W_mean"], "same_result": res_r == result})
    per_rank = None
    if world > 1:
        my_elapsed = elapsed
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # per-rank wall and stage times (and each GPU's clock and power), so that an imbalance between the shards is visible in the line
        mine = {"rank": rank, "device": local_rank, "ms_per_step": my_elapsed / args.steps * 1e3,
                "stage_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
                "clock_MHz_mean": tele["clock_MHz_mean"], "power_W_mean": tele["power_W_mean"]}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
    elif c_sharded:
        per_rank = ctx.shard_timings()
    # everything the line needs from the headline context is read NOW: the secondary measurements below close it to hand its
    # HBM back (round 3 queried it after the close at N = 8 and lost the line: VERDICT r3 weak #3)
    ctx_te_path = bool(ctx.query("twisted_edwards"))
    ctx_rccl = bool(ctx.query("rccl_exchanges")) if c_sharded else False
    geom = {k: ctx.query(k) for k in ("bucket_windows", "l1_bits", "l1_bins", "group_passes")} if not c_sharded else None
    g2_paired = ctx.query("g2_paired") if (cid >= 2 and not c_sharded) else None
    ctx_levels = ctx.query("table_levels")
    # mixed additions of one accumulate launch = non-zero digits (the entry capacity windows * n counts windows that stay (almost) empty:
    # the top window of an anchored plan); the anchored window itself, for the record
    sorted_entries = ctx.query("sorted_entries") if not c_sharded else 0
    anchored_window = ctx.query("anchored_window") if not c_sharded else 0
    te_limb_bits = ctx.query("te_limb_bits")   # 29: the Edwards kernels run on 13 x 29-bit limbs (337 multiply-adds per product), 28: 14 x 28 (378)

    # measurements that borrow the headline context (its bases ARE the workload's), then its ONE close
    borrowed = {}
    if rank == 0 and single and args.extras and cid < 2:
        for name, fn in (("host_scalars", measure_host_scalars), ("assume_subgroup", measure_assume_subgroup)):
            try:
                borrowed[name] = fn(E, ctx, result)
            except Exception as e:   # never lose the headline line to a secondary measurement
                borrowed[name] = {"error": repr(e)}
    ctx.close()
    ctx = None

    weak_point = None
    if weak_secondary is not None:
        # the weak-scaling point (2^26 pairs per GPU, what N = 1, 2, 4 measure) next to the configs[3] headline
        ctxw = None
        try:
            nw = 1 << weak_secondary
            if c_sharded:
                ctxw = ea.MultiScalarMultContext(args.curve, devices=devs)
                ctxw.set_bases(tile.repeat(nw // distinct, 1).repeat(args.gpus, 1))
                scw = torch.cat([uniform_scalars(nw, R377_TOP, device, seed=1234 + g) for g in range(args.gpus)])
            else:
                ctxw = ea.MultiScalarMultContext(args.curve, device=local_rank)
                ctxw.set_bases(tile.repeat(nw // distinct, 1).contiguous())
                scw = uniform_scalars(nw, R377_TOP, device, seed=1234 + rank)

            def stepw():
                partial = ctxw.run(scw)[0]
                if world > 1:
                    return ea.fold_partials(ea.all_gather_partials(partial, device=coll_device), args.curve)
                return partial

            stepw()
            fence()
            tw = time.perf_counter()
            for _ in range(args.steps):
                stepw()
            fence()
            tw = time.perf_counter() - tw
            if world > 1:
                tmax = torch.tensor([tw], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                tw = float(tmax.item())
            weak_point = {"workload": f"{args.curve} MSM, 2^{weak_secondary} pairs per GPU (weak scaling from the N = 1 line)",
                          "value": nw * n_ranks * args.steps / tw, "unit": "pairs/s", "ms_per_step": tw / args.steps * 1e3, "scaling": "weak"}
            del scw
        except Exception as e:
            weak_point = {"error": repr(e)}
        finally:
            if ctxw is not None:
                ctxw.close()

    if rank == 0:
        pairs_per_step = n * world * (args.gpus if c_sharded else 1)
        value = pairs_per_step * args.steps / elapsed
        kern_s = (acc_ms / max(acc_launches, 1)) * 1e-3
        pairs_per_launch = n * args.steps / max(acc_launches, 1)
        achieved = BYTES_PER_PAIR[cid] * pairs_per_launch / kern_s / 1e9
        traffic, traffic_raw, traffic_from, valu, pmc_stages = quoted_pmc(E, tm, ctx_te_path, total_npow)
        # the integer roofline: lane-level v_mad_u64_u32 per second in the dominant kernel against the measured
        # issue peak of 1024 SIMDs x 64 lanes / 4.3 cycles at the nominal 2.4 GHz (profiles/r01_ubench_valu_*.txt).
        # v_mad_u64_u32 per mixed addition: a property of the formulas, pinned on the generated ISA by tests/test_isa.py -- 7 multiplications
        # of 337 (Edwards on 13 x 29 limbs; 378 on 14 x 28); 6M + 2S + one fused dual product (XYZZ over Fp); G2 with two lanes per point: 10 fused dual products per lane
        # (574 each with the p0 = 1 shortcut of BLS12-377, 588 for BLS12-381), i.e. 20 per addition; one lane per point: 8 x 2 + 2 x (1 + 2/3).
        paired = bool(g2_paired and (g2_paired & 1))
        mads_per_add = {0: ((2359 if te_limb_bits == 29 else 2646) if ctx_te_path else 3416), 1: 3542, 2: 11480 if paired else 11088, 3: 11760 if paired else 11368}[cid]
        adds_per_launch = sorted_entries or tm["entries"]   # one mixed addition per sorted entry
        mad_rate = mads_per_add * adds_per_launch / kern_s
        mad_peak = MAD_PEAK
        ms_all = sorted(s["ms_per_step"] for s in samples)
        step_ms = elapsed / args.steps * 1e3
        at_ref = None
        if tele["clock_MHz_mean"]:
            # what this step would take at the reference clock if only the (VALU-bound, hence clock-proportional) accumulation scaled
            acc_step = stage_ms["accumulate"] / args.steps
            at_ref = step_ms - acc_step + acc_step * (tele["clock_MHz_mean"] / 1e3) / REF_CLOCK_GHZ
        out = {
            "metric": {0: "BLS12-377 G1", 1: "BLS12-381 G1", 2: "BLS12-377 G2", 3: "BLS12-381 G2"}[cid] + " MSM point-scalar pairs/s",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": args.gpus if c_sharded else world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_ms,
            "ms_per_2^26_pairs": step_ms * (1 << 26) / n,
            "ms_per_step_at_1p95GHz": at_ref,
            "headline_samples": {"ms_per_step": [s["ms_per_step"] for s in samples], "median": ms_all[len(ms_all) // 2], "min": ms_all[0], "max": ms_all[-1],
                                 "per_sample": samples,
                                 "what": f"the same {args.steps}-step timed loop taken {len(samples)} times back to back (the first is `value`), each with the "
                                         "GPU's mean shader clock and socket power sampled while it ran"},
            "higher_is_better": True,
            "scaling": "weak" if not total_npow else "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "dtype_detail": ("13 x 29-bit limbs in u32 lanes, Montgomery radix 2^406 (twisted-Edwards kernels of BLS12-377 G1); " if (ctx_te_path and te_limb_bits == 29) else "")
                            + "14 x 28-bit limbs in u32 lanes, Montgomery radix 2^392" + (" everywhere else" if (ctx_te_path and te_limb_bits == 29) else "")
                            + "; products accumulated in u64 (v_mad_u64_u32)",
            "data": "synthetic: 2^15 distinct subgroup points replicated (reference generator shape), uniform scalars < r",
            "config": {"workload": (f"{args.curve} MSM, 2^{total_npow} pairs sharded over {n_ranks} GPU(s) (2^{args.npow} per GPU; "
                                    + ("BASELINE.json configs[3]" if (total_npow == 28 and n_ranks == 8) else "the shape of BASELINE.json configs[3] at another size")
                                    + "), bases+scalars resident in HBM" if total_npow else
                                    f"{args.curve} MSM, 2^{args.npow} pairs per GPU, bases+scalars resident in HBM"),
                       "pairs_per_gpu": n, "window_bits": tm["window_bits"], "windows": tm["windows"],
                       "anchored_window": (anchored_window - 1) if anchored_window else None,
                       "mixed_additions_per_launch": adds_per_launch,
                       "lane_entries": tm["lane_entries"], "precompute": args.precompute, "table_levels": ctx_levels,
                       "group_law": "extended twisted Edwards (7M mixed add)" if ctx_te_path else "XYZZ (8M+2S mixed add)",
                       "g2_paired": g2_paired,
                       "init_s": t_init,
                       "clock_MHz_mean": tele["clock_MHz_mean"], "clock_MHz_min": tele["clock_MHz_min"], "clock_MHz_max": tele["clock_MHz_max"],
                       "power_W_mean": tele["power_W_mean"], "power_W_max": tele["power_W_max"], "telemetry_samples": tele["samples"],
                       "telemetry_source": telemetry.describe(),
                       "parallelism": (f"one process, {args.gpus} shards behind the C ABI (mi355_msm_create_sharded), "
                                       f"{'RCCL all-gather' if ctx_rccl else 'host fold'} of {args.gpus} partial points" if c_sharded else
                                       f"{world} disjoint base/scalar slices + all-gather of {world} partial points")},
            "stage_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
            "stage_roofline": stage_roofline(n, tm, {k: v / args.steps for k, v in stage_ms.items()}, cid, geom, pmc_stages) if geom else None,
            "per_rank": per_rank,
            "weak_scaling_point": weak_point,
            "roofline": {"bound": "hbm", "kernel": "k_accumulate_glds", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "peak_measured": max(probe.get("copy_GBps", 0), probe.get("read_GBps", 0)) if probe and "error" not in probe else None,
                         "peak_probe": probe,
                         "traffic": traffic, "traffic_raw_counter_bytes": traffic_raw, "traffic_from": traffic_from,
                         "kernel_ms": kern_s * 1e3, "algorithmic_bytes_per_launch": BYTES_PER_PAIR[cid] * pairs_per_launch,
                         "valu": valu,
                         "integer": {"mads_per_mixed_add": mads_per_add, "lane_mads_per_s": mad_rate, "peak_lane_mads_per_s": mad_peak,
                                     "frac": mad_rate / mad_peak,
                                     "peak_is": "v_mad_u64_u32 issue limit at the nominal 2.4 GHz; the kernel runs power-limited near 1.9 GHz"},
                         "note": "integer-VALU-bound path (no MFMA): the binding resource is VALU issue at the power-limited clock (DESIGN.md sections 2, 7)"},
        }
        if single and args.extras and cid < 2:
            # `value` above keeps the scalars in HBM (the brief's rule); these are the PCIe-inclusive figures, measured in this same run
            extras = {}
            for k in ("host_scalars", "assume_subgroup"):
                if k in borrowed:
                    extras[k] = borrowed[k]
            for name, fn in (("xyzz", measure_xyzz if cid == 0 else None), ("stateless", measure_stateless)):
                if fn is None:
                    continue
                try:
                    extras.update(fn(E, result))
                except Exception as e:
                    extras[name] = {"error": repr(e)}
            out["survey_8d_metrics"] = extras
            # SURVEY.md 8(d)'s primary figure at the TOP LEVEL.  `value` itself stays the device-resident rate: the
            # task's measurement rule fixes it ("inputs already resident in HBM when the timed region starts ... the PCIe-inclusive
            # rate ... is never `value`"); this object is the same MSM as the reference's timed closure sees it
            # (P1A combined-top-solutions/benches/msm.rs:21,27-35: bases resident, scalars handed over in host memory).
            hs = extras.get("host_scalars", {})
            if "one_batch_ms" in hs:
                out["host_scalars_ms_per_msm"] = {
                    "pageable": hs["one_batch_ms"]["pageable"], "pinned": hs["one_batch_ms"]["pinned"],
                    "four_batches_pageable": hs.get("four_batches_ms", {}).get("pageable"),
                    "device_resident_ms": step_ms,
                    "same_result": hs.get("same_result_as_device_scalars"),
                    "what": "SURVEY 8(d) primary metric: one mi355_msm_run() with the 2^%d scalars in HOST memory (PCIe upload inside the call), bases resident; "
                            "`value` / `ms_per_step` are the device-resident figure the measurement rule asks for" % args.npow}
        if single and args.extras and cid == 0 and args.npow == 26:
            out["secondary_configs"] = measure_secondary_configs(E)
            out["small_input_latency"] = measure_small_latency(E)
        if single and args.cpu_sample_pow > 0:
            try:
                out["cpu_baseline"] = measure_cpu_baseline(E, result)
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        if single and args.also_precompute and not args.precompute and cid < 2:
            try:
                out["with_precomputed_tables"] = measure_tables(E, result, bool(args.extras))
                table = out["with_precomputed_tables"]["by_table_levels"]
                ok_rows = {k: v for k, v in table.items() if "error" not in v and "four_batches_from_pageable_host_ms" in v}
                hs = out.get("survey_8d_metrics", {}).get("host_scalars", {})
                if ok_rows and "four_batches_ms" in hs:
                    best = min(ok_rows, key=lambda k: ok_rows[k]["four_batches_from_pageable_host_ms"])
                    hs["four_batches_ms"]["precompute"] = {k: v["four_batches_from_pageable_host_ms"] for k, v in ok_rows.items()}
                    hs["four_batches_ms"]["precompute_best_table_levels"] = best
            except Exception as e:
                out["with_precomputed_tables"] = {"error": str(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
