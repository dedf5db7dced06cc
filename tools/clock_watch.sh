#!/bin/bash
# Sample GPU clock and power while a command runs: tools/clock_watch.sh <out.txt> -- <cmd...>
out=$1; shift; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr '\n' ' ' >> $out; echo >> $out; sleep 0.2; done ) &
wp=$!
"$@"
rc=$?
kill $wp
exit $rc
