"""GPU: bench.py in the exact command shapes the driver launches, rehearsed on the one GPU of the test box.

For N > 1 the driver runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
bench.py --gpus N --steps K --warmup W`, one rank per GPU over RCCL.  Here the N ranks share device 0 and exchange their
144-byte partials over gloo (`--backend gloo`); everything else -- the default workload choice per N (N = 8: the shape of BASELINE
configs[3], four times the single-GPU problem over eight shards, plus the weak-scaling point in the same run), the per-rank
report, the max-over-ranks timing, the one JSON line on rank 0 -- is the code the driver will execute.  Round 3 shipped an N = 8
line that raised before printing (a query on a closed context); this test is what would have caught it.

The lines are kept under gpurun_out/ (copied to profiles/r05_bench_rehearsal_*.json).
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

NPOW = 17


def _line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}:\n{stdout[-2000:]}"
    return json.loads(lines[0])


def _keep(name: str, line: dict) -> None:
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"r05_bench_rehearsal_{name}.json"), "w") as f:
            json.dump(line, f, indent=1)
    except OSError:
        pass


def _check_common(line: dict, n: int) -> None:
    assert line["n_gpus"] == n and line["steps"] == 2 and line["warmup"] == 1
    assert line["unit"] == "pairs/s" and line["value"] > 0 and line["higher_is_better"] is True
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["frac"] > 0
    assert "model" not in line["config"] and "workload" in line["config"]
    assert isinstance(line["per_rank"], list) and len(line["per_rank"]) == n


@pytest.mark.parametrize("n", [2, 4, 8])
def test_driver_command_shape_multi_rank(built, n):
    port = 29900 + (os.getpid() + n) % 500
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", str(n), "--backend", "gloo", "--npow", str(NPOW), "--steps", "2", "--warmup", "1",
           "--cpu-sample-pow", "0"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _line(r.stdout)
    _keep(f"torchrun_n{n}", line)
    _check_common(line, n)
    assert sorted(p["rank"] for p in line["per_rank"]) == list(range(n))
    if n == 8:
        # the default --total-npow logic fired: 2^(npow+2) pairs over 8 shards, and the weak-scaling point ran in the same process
        assert line["scaling"] == "strong"
        assert "configs[3]" in line["config"]["workload"] and f"2^{NPOW + 2} pairs" in line["config"]["workload"]
        assert line["config"]["pairs_per_gpu"] == 1 << (NPOW - 1)
        wp = line["weak_scaling_point"]
        assert wp and "error" not in wp, wp
        assert wp["value"] > 0 and f"2^{NPOW} pairs per GPU" in wp["workload"]
    else:
        assert line["scaling"] == "weak" and line["weak_scaling_point"] is None
        assert line["config"]["pairs_per_gpu"] == 1 << NPOW


def test_driver_command_shape_single_process_sharded(built):
    """`python bench.py --gpus 8` without a launcher: one process, eight shards behind the C ABI (here all on device 0)."""
    cmd = [sys.executable, "bench.py", "--gpus", "8", "--logical-shards", "1", "--npow", str(NPOW), "--steps", "2", "--warmup", "1",
           "--cpu-sample-pow", "0"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _line(r.stdout)
    _keep("single_process_n8", line)
    _check_common(line, 8)
    assert "configs[3]" in line["config"]["workload"]
    wp = line["weak_scaling_point"]
    assert wp and "error" not in wp, wp
    assert "host fold" in line["config"]["parallelism"] or "RCCL" in line["config"]["parallelism"]


def test_default_single_gpu_line_small(built):
    """The N = 1 shape with every secondary measurement switched on, at a size that finishes in seconds: the line must carry
    `roofline`, `cpu_baseline`, the host-scalar figures and the stateless call without an `error` member anywhere."""
    cmd = [sys.executable, "bench.py", "--npow", "18", "--steps", "2", "--warmup", "1", "--cpu-sample-pow", "14", "--also-precompute", "1"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _line(r.stdout)
    _keep("n1_small", line)
    assert line["n_gpus"] == 1 and line["per_rank"] is None
    assert line["cpu_baseline"]["gpu_matches_cpu_on_sample"] is True
    assert "error" not in json.dumps(line["survey_8d_metrics"]), line["survey_8d_metrics"]
    assert "error" not in line["with_precomputed_tables"], line["with_precomputed_tables"]
