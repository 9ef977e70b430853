#!/bin/bash
# Runs ON the GPU box (via gpurun): the default bench, its rocprofv3 kernel trace and three PMC
# passes (separate runs, counters only with --kernel-trace, as the pool requires), all under
# gpurun_out/<tag>/.  Usage: tools/profile_on_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
# traced runs: the timed region only (no CPU baseline, no ensemble leg -- its concurrent chains launch the same
# kernels on other data sets and would be averaged into the per-kernel statistics)
B="python3 bench.py --steps 150 --warmup 10 --no-cpu-baseline --experiments 0 $@"
timeout -k 10 600 python3 bench.py "$@" > $OUT/bench.json.log 2>&1 || { tail -5 $OUT/bench.json.log; exit 1; }
tail -1 $OUT/bench.json.log > $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || exit 5
python3 tools/summarize_rocprof.py stats $OUT/trace $OUT/kernel_stats_summary.csv > /dev/null
python3 tools/summarize_rocprof.py timeline $OUT/trace $OUT/timeline_summary.csv > /dev/null
python3 tools/summarize_rocprof.py pmc $OUT/pmc_summary.csv $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write > /dev/null
# keep the summaries, drop the raw per-dispatch traces (gpurun_out/ is capped at 64 MiB)
rm -rf $OUT/trace $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write
cut -c1-400 $OUT/bench.json
cat $OUT/kernel_stats_summary.csv | head -8
cat $OUT/timeline_summary.csv | head -8
grep -E "fill_" $OUT/pmc_summary.csv
