#!/bin/bash
# Run on the GPU box (via gpurun): kernel-time stats and HBM-traffic counters for the bench workload.
# Counters are collected in their own passes (kernel-trace only), as the pool requires.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/prof}
rm -rf $OUT; mkdir -p $OUT
NPOW=${NPOW:-26}
EXTRA=${EXTRA:-}   # e.g. EXTRA="--curve bls12_377_g2" NPOW=24 OUT=gpurun_out/prof_g2
BENCH="python bench.py --steps 3 --warmup 1 --repeat 0 --hbm-probe 0 --cpu-sample-pow 0 --extras 0 --also-precompute 0 --npow $NPOW $EXTRA"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats_bench.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  TAG=$(echo $C | tr ' ' '_')
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$TAG -- python bench.py --steps 1 --warmup 0 --repeat 0 --hbm-probe 0 --cpu-sample-pow 0 --extras 0 --also-precompute 0 --npow $NPOW $EXTRA > $OUT/pmc_$TAG.json 2> $OUT/pmc_$TAG.err
done
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
# keep only the small artefacts (gpurun_out is capped at 64 MiB)
find $OUT -name "*.csv" -size +8M -delete
cat $OUT/summary.txt
