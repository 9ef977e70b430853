#!/bin/bash
# Runs ON the GPU box: rocprofv3 kernel statistics of the ensemble leg alone -- ONE lockstep set (so that the set's
# kernels do not overlap with another set's and the per-kernel durations are their own), 2 chains per fill, and the
# same with 4.  Usage: tools/profile_lockstep.sh <tag>
set -o pipefail
TAG=$1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
for L in 2 4; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$L -- python3 bench.py --steps 20 --warmup 5 \
    --also none --no-cpu-baseline --prewarm 50 --experiments $L --exp-steps 4000 --exp-lockstep $L --exp-sets 1 --exp-concurrent 1 \
    > $OUT/trace_$L.log 2>&1 || exit 2
  python3 tools/summarize_rocprof.py stats $OUT/trace_$L $OUT/lockstep_${L}_kernel_stats.csv > /dev/null
  rm -rf $OUT/trace_$L
  head -12 $OUT/lockstep_${L}_kernel_stats.csv
done
