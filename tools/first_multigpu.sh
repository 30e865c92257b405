#!/bin/bash
# The order to run things in on the FIRST box with >= 2 MI355X (no such box has been available to any round so far: the xGMI hop, RCCL
# with more than one rank and the cross-device copies have never executed -- DESIGN.md section 9).  Every step says what a failure means
# and which file to look at.  Run from the repository root:   bash tools/first_multigpu.sh [N]      (N = GPUs to use, default all)
# Output: gpurun_out/first_multigpu/*.txt -- copy into profiles/ what you want kept.
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0          # the host driver only supports dmabuf IPC: without it RCCL fails in hipIpcGetMemHandle
export MASTER_ADDR=127.0.0.1
OUT=gpurun_out/first_multigpu
mkdir -p $OUT
NDEV=$(python -c "import torch; print(torch.cuda.device_count())")
N=${1:-$NDEV}
echo "visible devices: $NDEV, using $N" | tee $OUT/00_devices.txt
if [ "$NDEV" -lt 2 ]; then echo "needs >= 2 GPUs"; exit 1; fi

step() {   # step <name> <timeout s> <command...>
  local name=$1 to=$2; shift 2
  echo "== $name: $*"
  if timeout "$to" "$@" > $OUT/$name.txt 2>&1; then echo "   ok"; else echo "   FAILED (rc $?) -- see $OUT/$name.txt"; FAILED="$FAILED $name"; fi
}
FAILED=""

# 1. The single-GPU suite's sharded tests first (logical shards + the forced peer-staging branches): if THESE fail the tree is broken,
#    not the node.
step 01_sharded_one_gpu 900 python -m pytest tests/test_gpu_sharded.py -x -q

# 2. One process, all devices, behind the C ABI (csrc/msm_sharded.hpp): bases staged from device 0 to every shard
#    (sharded_set_bases peer branch), scalars pulled per batch (sharded_run peer branch), combine = 2 = RCCL all-gather REQUIRED.
#      "ncclCommInitAll failed"          -> RCCL / xGMI bring-up (check rocm-smi --showtopo, HSA_ENABLE_IPC_MODE_LEGACY=0)
#      "partials that differ"            -> the all-gather moved wrong bytes: a link problem, not an MSM problem
#      result != oracle with combine = 1 -> a cross-device copy is wrong (hipMemcpyDefault peer path): run step 1's
#                                           test_peer_staging_branches_on_logical_shards to separate logic from transport
step 02_multidevice_tests 1800 python -m pytest tests/test_gpu_multidevice.py -x -q -rA

# 3. The driver's own command shapes, small first (seconds), then full size.  One JSON line each; `per_rank` shows imbalance.
#      a hang at init_process_group      -> rendezvous (MASTER_ADDR must be 127.0.0.1) or RCCL IPC (the env var above)
#      value(N) / value(1) well below N  -> look at per_rank[*].stage_ms_per_step: the MSM itself shards perfectly, so a slow rank is
#                                           a slow GPU (power cap, clock) or its PCIe link when scalars come from the host
step 03_bench_n1_small 600 python bench.py --only headline --npow 20
for G in 2 4 8; do
  [ "$G" -le "$N" ] || continue
  step 04_bench_torchrun_n${G}_small 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port $((29500 + G)) \
       bench.py --gpus $G --steps 3 --warmup 1 --npow 20
  step 05_bench_single_process_n${G}_small 900 python bench.py --gpus $G --steps 3 --warmup 1 --npow 20
done
step 06_bench_n1 900 python bench.py --only headline
for G in 2 4 8; do
  [ "$G" -le "$N" ] || continue
  step 07_bench_torchrun_n${G} 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port $((29600 + G)) \
       bench.py --gpus $G --steps 5 --warmup 1
  step 08_bench_single_process_n${G} 1800 python bench.py --gpus $G --steps 5 --warmup 1
done

# 4. Scaling summary (the driver computes efficiency itself; this is for the person at the keyboard)
python - <<'EOF' | tee $OUT/99_summary.txt
import glob, json, os, re
rows = {}
for f in sorted(glob.glob("gpurun_out/first_multigpu/0[678]_bench_*.txt")):
    for line in open(f):
        if line.startswith("{"):
            j = json.loads(line)
            rows[os.path.basename(f)] = (j["n_gpus"], j["value"], j["ms_per_step"], j["config"]["workload"])
base = next((v[1] for k, v in rows.items() if v[0] == 1), None)
for k, (n, v, ms, w) in rows.items():
    print("%-40s N=%d  %.3e pairs/s  %.1f ms/step  x%.2f  %s" % (k, n, v, ms, v / base if base else float("nan"), w))
EOF
# 5. measured against what the single-GPU numbers predict (profiles/r06_scale_model.json; tools/scale_model.py)
python tools/scale_model.py --compare $OUT | tee $OUT/98_vs_model.txt
[ -z "$FAILED" ] && echo "all steps passed" || echo "failed steps:$FAILED"
