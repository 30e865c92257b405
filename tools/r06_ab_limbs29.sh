#!/bin/bash
# On the GPU box: the 14 x 28 vs 13 x 29 A/B of the twisted-Edwards hot path.
#   1. tools/ubench_madd: the mixed addition in isolation, both limb shapes in one process;
#   2. the element-wise device tests of the new shape, the Edwards tests and the BLS12-377 parity tests;
#   3. the whole engine, variants staged by tools/build_variant.sh (te28 = -DMSM_TE_LIMBS29=0), interleaved on this box.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r06_ab_limbs29
mkdir -p $O
tools/ubench_madd 4000 > $O/ubench_madd.txt 2>&1
python -m pytest tests/test_gpu_limbs29.py tests/test_gpu_devtest.py tests/test_gpu_te.py -x -q 2>&1 | tail -5 > $O/pytest_new.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fold.py tests/test_gpu_carry.py -x -q 2>&1 | tail -5 > $O/pytest_parity.txt
bash tools/ab_bench.sh 3 > $O/ab_engine.txt 2>&1
cat $O/ubench_madd.txt $O/pytest_new.txt $O/pytest_parity.txt $O/ab_engine.txt
