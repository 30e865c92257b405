#!/bin/bash
# On the GPU box: everything the round's evidence files are made from, on the tree as it is.
#   calibration of FETCH_SIZE / WRITE_SIZE, rocprofv3 stats + PMC passes for the three bench curves, the GPU suite, the bench line.
set -u
cd "$(dirname "$0")/.."
bash tools/calib_fetch_run.sh > /dev/null 2>&1
cp gpurun_out/calib/summary.txt gpurun_out/r03_calib_fetch.txt
OUT=gpurun_out/prof bash tools/profile_gpu.sh > /dev/null 2>&1
EXTRA="--curve bls12_381_g1" OUT=gpurun_out/prof_381 bash tools/profile_gpu.sh > /dev/null 2>&1
EXTRA="--curve bls12_377_g2" NPOW=24 OUT=gpurun_out/prof_g2 bash tools/profile_gpu.sh > /dev/null 2>&1
for t in "prof 377 " "prof_381 381 _381" "prof_g2 g2 _g2"; do
  set -- $t
  python tools/make_pmc_json.py gpurun_out/$1 gpurun_out/r03_pmc_k_accumulate${3:-}.json $2 gpurun_out/r03_calib_fetch.txt > /dev/null 2>&1
  cp gpurun_out/$1/summary.txt gpurun_out/r03_rocprof_summary${3:-}.txt
  # bench.py quotes roofline.traffic from profiles/ when the counters were measured on the kernel sources it runs: these were
  cp gpurun_out/r03_pmc_k_accumulate${3:-}.json profiles/
done
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 > gpurun_out/r03_pytest_gpu.txt
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
python bench.py --curve bls12_381_g1 --cpu-sample-pow 22 > gpurun_out/r03_bench_381.json 2>> gpurun_out/r03_bench.err
python bench.py --curve bls12_377_g2 --npow 24 --cpu-sample-pow 20 > gpurun_out/r03_bench_g2.json 2>> gpurun_out/r03_bench.err
find gpurun_out -name "*.csv" -size +2M -delete
cat gpurun_out/r03_calib_fetch.txt; tail -3 gpurun_out/r03_pytest_gpu.txt
