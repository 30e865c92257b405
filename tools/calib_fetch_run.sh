#!/bin/bash
# On the GPU box: the two counter passes over tools/calib_fetch and the ratio table (profiles/r03_calib_fetch.txt).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
tools/calib_fetch > $OUT/calib_expected.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- tools/calib_fetch > /dev/null 2> $OUT/fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- tools/calib_fetch > /dev/null 2> $OUT/write.err
python tools/calib_fetch_summary.py $OUT | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +4M -delete
