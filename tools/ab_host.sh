cd "$(dirname "$0")/.."
LIB=2022-entries_amd/libmi355msm.so
cp $LIB /tmp/keep.so
for r in 1 2; do
  for v in 2022-entries_amd/build/variants/*.so; do
    cp $v $LIB
    echo -n "$(basename $v .so) r$r: "
    python tools/host_cold_probe.py 2>/dev/null | grep "warm\|pinned" | awk '{printf "%s ", $(NF-2)}'; echo
  done
done
cp /tmp/keep.so $LIB
