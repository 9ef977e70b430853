#!/bin/bash
# Runs ON the GPU box: the unchanged call sequence (bench_cpp --reference-form: S x EvalAsync, S x EvalFinished,
# nll_event_chunks, finish_nll_jump_pick_combo per step) under the knobs of the deferred batch -- which stream the
# batch goes to, how EvalFinished waits -- each timed plain, and the default once under rocprofv3 --kernel-trace for the
# device timeline (kernel durations and the idle gaps between them).  Usage: tools/dropin_study.sh <tag>
set -o pipefail
TAG=${1:-dropin}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
B="./tests/cpp/bench_cpp --reference-form --steps 3000"
for rep in 1 2; do
  for cfg in "default" "SXMC_DEFER_STREAM=own" "SXMC_FINISH_SPIN_US=0" "SXMC_DEFER_STREAM=own SXMC_FINISH_SPIN_US=0" "SXMC_DEFER_EVAL=0"; do
    if [ "$cfg" = "default" ]; then line=$($B 2>>$OUT/err.log); else line=$(env $cfg $B 2>>$OUT/err.log); fi
    echo "$cfg: $(echo $line | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["steps_per_sec_stepping"], "steps/s,", r["deferred_launches"], "launches")')" | tee -a $OUT/knobs.log
  done
done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- ./tests/cpp/bench_cpp --reference-form --steps 600 > $OUT/trace.log 2>&1 || exit 2
python3 tools/summarize_rocprof.py timeline $OUT/trace $OUT/timeline.csv 0.6
rm -rf $OUT/trace
