#!/bin/bash
# Runs ON the GPU box: kernel-trace statistics of one bench run, summary printed.  Usage: tools/quick_stats.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 20 --prewarm 100 --no-cpu-baseline --experiments 0 "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 2; }
python3 tools/summarize_rocprof.py stats $OUT/trace $OUT/kernel_stats_summary.csv > /dev/null
rm -rf $OUT/trace
head -8 $OUT/kernel_stats_summary.csv
tail -1 $OUT/trace.log | cut -c1-200
