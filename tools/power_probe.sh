#!/bin/bash
# Run on the GPU box: sample socket power and shader clock while the bench (or any command) runs.
# usage: tools/power_probe.sh <outfile> -- <command...>
OUT=$1; shift; shift
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' ' ; echo; sleep 0.15; done ) > "$OUT" &
PROBE=$!
"$@"
RC=$?
kill $PROBE
exit $RC
