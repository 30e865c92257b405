#!/bin/bash
# On the GPU box: everything the round's evidence files are made from, on the tree as it is.
#   rocprofv3 stats + PMC passes for EVERY product curve (FETCH_SIZE / WRITE_SIZE corrected with the TRACKED calibration
#   profiles/r03_calib_fetch.txt; per-stage counters of the grouping and bucket-reduction kernels in the same passes), the GPU suite,
#   the bench lines.  PARTS="prof tests bench" selects (default all).
set -u
cd "$(dirname "$0")/.."
R=r06
PARTS=${PARTS:-prof tests bench}
if [[ " $PARTS " == *" prof "* ]]; then
  OUT=gpurun_out/prof bash tools/profile_gpu.sh > /dev/null 2>&1
  EXTRA="--curve bls12_381_g1" OUT=gpurun_out/prof_381 bash tools/profile_gpu.sh > /dev/null 2>&1
  EXTRA="--curve bls12_377_g2" NPOW=24 OUT=gpurun_out/prof_g2 bash tools/profile_gpu.sh > /dev/null 2>&1
  EXTRA="--curve bls12_381_g2" NPOW=24 OUT=gpurun_out/prof_381g2 bash tools/profile_gpu.sh > /dev/null 2>&1
  for t in "prof 377 " "prof_381 381 _381" "prof_g2 g2 _g2" "prof_381g2 381g2 _381g2"; do
    set -- $t
    python tools/make_pmc_json.py gpurun_out/$1 gpurun_out/${R}_pmc_k_accumulate${3:-}.json $2 profiles/r03_calib_fetch.txt > /dev/null 2>&1
    cp gpurun_out/$1/summary.txt gpurun_out/${R}_rocprof_summary${3:-}.txt
    # bench.py quotes roofline.traffic from profiles/ when the counters were measured on the kernel sources AND the plan it runs: these were
    cp gpurun_out/${R}_pmc_k_accumulate${3:-}.json profiles/
  done
  cp $(find gpurun_out/prof/stats -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_kernel_stats.csv 2>/dev/null
  find gpurun_out -name "*.csv" -size +2M -delete
fi
if [[ " $PARTS " == *" tests "* ]]; then
  python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/${R}_pytest_gpu_full.txt
  # (the whole report when something failed, the summary otherwise)
  if grep -q "failed" gpurun_out/${R}_pytest_gpu_full.txt; then tail -120 gpurun_out/${R}_pytest_gpu_full.txt > gpurun_out/${R}_pytest_gpu.txt; else tail -6 gpurun_out/${R}_pytest_gpu_full.txt > gpurun_out/${R}_pytest_gpu.txt; fi
  rm -f gpurun_out/${R}_pytest_gpu_full.txt
fi
if [[ " $PARTS " == *" bench "* ]]; then
  python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
  python bench.py --curve bls12_381_g1 --cpu-sample-pow 22 --also-precompute 0 > gpurun_out/${R}_bench_381.json 2>> gpurun_out/${R}_bench.err
  python bench.py --curve bls12_377_g2 --npow 24 --cpu-sample-pow 20 > gpurun_out/${R}_bench_g2.json 2>> gpurun_out/${R}_bench.err
  python bench.py --curve bls12_381_g2 --npow 24 --cpu-sample-pow 20 > gpurun_out/${R}_bench_381g2.json 2>> gpurun_out/${R}_bench.err
  for i in 1 2 3; do python bench.py --only headline 2>> gpurun_out/${R}_bench.err; done > gpurun_out/${R}_bench_headline_x3.json
fi
tail -3 gpurun_out/${R}_pytest_gpu.txt 2>/dev/null; tail -c 600 gpurun_out/${R}_bench.json 2>/dev/null
