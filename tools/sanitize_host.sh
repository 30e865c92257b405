#!/bin/bash
# Host side under sanitizers ("race detection / sanitizers"):
#   1. libmsm_hosttest.so rebuilt with -fsanitize=address,undefined (the fp28 / curve / te / host_fold64 templates, the op table of
#      devtest_ops.hpp, the mulx/adx assembly of the 64-bit fold) and the CPU tests that drive it run under it;
#   2. the yrrid shim's hex readers (C, file parsing) built the same way and driven by their test;
#   3. tests/tsan_pipeline.cpp: the stateless pipeline's staging threads / slot ring / slice events and the per-shard thread
#      fan-out (csrc/host_pipeline.hpp) against a fake asynchronous copy engine, under -fsanitize=thread.
# Writes profiles/r03_sanitizers.txt.  No GPU needed.
set -u
cd "$(dirname "$0")/.."
PKG=2022-entries_amd
OUT=profiles/r03_sanitizers.txt
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
{
echo "# tools/sanitize_host.sh  ($(gcc --version | head -1))"
echo "== 1. libmsm_hosttest.so with -fsanitize=address,undefined -fno-sanitize-recover=undefined"
cp $PKG/libmsm_hosttest.so /tmp/hosttest_keep.so
g++ -O1 -g -std=c++17 -shared -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -o $PKG/libmsm_hosttest.so $PKG/csrc/host_test_api.cpp || { echo "sanitized build FAILED"; exit 1; }
nm -D $PKG/libmsm_hosttest.so | grep -c __asan_ | sed "s/^/__asan_ symbols referenced by the library under test: /"
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  timeout 3000 python -m pytest tests/test_field_host.py tests/test_te_host.py tests/test_fold64_host.py tests/test_devtest_host.py -q -p no:cacheprovider 2>&1 | tail -4
cp /tmp/hosttest_keep.so $PKG/libmsm_hosttest.so
echo "== 2. hex readers of the yrrid shim with -fsanitize=address,undefined"
cp $PKG/libmi355msm_yrrid_377.so /tmp/yrrid_keep.so
gcc -O1 -g -shared -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -DFEATURE_BLS12_377 -o $PKG/libmi355msm_yrrid_377.so \
  $PKG/csrc/shims/yrrid_context.c -L$PKG -lmi355msm -Wl,-rpath,'$ORIGIN'
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1 \
  timeout 600 python -m pytest tests/test_hex_readers.py -q -p no:cacheprovider 2>&1 | tail -3
cp /tmp/yrrid_keep.so $PKG/libmi355msm_yrrid_377.so
echo "== 3. staging pipeline + shard fan-out under -fsanitize=thread (tests/tsan_pipeline.cpp)"
g++ -O1 -g -std=c++20 -fsanitize=thread -pthread -I$PKG/csrc -o /tmp/tsan_pipeline tests/tsan_pipeline.cpp && TSAN_OPTIONS=halt_on_error=1 /tmp/tsan_pipeline
echo "exit code $?"
echo "== 3b. the same harness under -fsanitize=address,undefined"
g++ -O1 -g -std=c++20 -fsanitize=address,undefined -fno-sanitize-recover=undefined -pthread -I$PKG/csrc -o /tmp/asan_pipeline tests/tsan_pipeline.cpp && /tmp/asan_pipeline
echo "exit code $?"
} 2>&1 | tee $OUT
