set -u
D=gpurun_out/r02final; mkdir -p $D
python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v amdgpu.ids > $D/pytest_gpu.txt; tail -3 $D/pytest_gpu.txt
bash tools/profile_gpu.sh > $D/rocprof_summary.txt 2>&1
python tools/make_pmc_json.py gpurun_out/prof > /dev/null 2>&1; cp profiles/r02_pmc_k_accumulate.json $D/pmc_k_accumulate.json
cp $(find gpurun_out/prof/stats -name "*kernel_stats.csv" | head -1) $D/kernel_stats.csv 2>/dev/null
OUT=gpurun_out/prof_g2 NPOW=24 EXTRA="--curve bls12_377_g2" bash tools/profile_gpu.sh > $D/rocprof_summary_g2.txt 2>&1
python tools/make_pmc_json.py gpurun_out/prof_g2 $D/pmc_k_accumulate_g2.json 13 24 20 "bls12_377_g2 npow=24 (c = 20, 13 windows): the k_accumulate_glds<SwLaw<Fp2El>> launch of a bench step, XYZZ over Fp2, one wave per SIMD" > /dev/null 2>&1
python bench.py 2>/dev/null | tail -1 > $D/bench.json
python bench.py --curve bls12_381_g1 --extras 0 2>/dev/null | tail -1 > $D/bench_381.json
python bench.py --curve bls12_377_g2 --npow 24 --extras 0 2>/dev/null | tail -1 > $D/bench_g2.json
python tools/size_sweep.py 2>/dev/null | grep "^2\^" > $D/size_sweep.txt
python tools/host_rate.py 2>/dev/null | grep -v amdgpu.ids > $D/host_rate.txt
python tools/fuzz_gpu.py 400 2>&1 | tail -3 > $D/fuzz.txt
rm -rf gpurun_out/prof gpurun_out/prof_g2
python - <<'PY'
import json
for f in ("bench","bench_381","bench_g2"):
    j=json.load(open("gpurun_out/r02final/%s.json"%f)); print(f, round(j["ms_per_step"],2), j["roofline"]["traffic"], j["roofline"].get("traffic_from"))
PY
cat $D/pmc_k_accumulate.json | python -c "import json,sys; j=json.load(sys.stdin); print(j['kernel_source_sha16'], j['derived'], j['kernel_ms_rocprof'])"
cat $D/pmc_k_accumulate_g2.json | python -c "import json,sys; j=json.load(sys.stdin); print(j['derived'], j['kernel_ms_rocprof'])"
