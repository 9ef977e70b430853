#!/bin/bash
# THE GATED STEP, an experiment on the measurement build ("one more look"): every step's fill on a
# second stream beside the previous step's step end, its prologue done before the proposal is written.  A/B on one box.
out=${1:-gpurun_out/gated_step_ab.log}
: > $out
export SXMC_HIP_LIB=$PWD/sxmc_amd/csrc/libsxmc_hip_measure.so
run() {
  label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --also none --experiments 0 --no-cpu-baseline --no-autotune --steps 1000 --launch 768,1 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-44s %8.1f evals/s  step %.2f us  fill %.2f us  accepted %.3f  %s' % ('$label', r['value'], 1e3*r['ms_per_step'], 1e3*f['avg_launch_ms'], r['config']['accepted_fraction_rank0'], r['config']['launch_plan'][-22:]))" >> $out
}
for k in 1 2 3; do
  run "768x1, 20 KB of LDS left free, one stream" SXMC_LDS_RESERVE=20480
  run "... gated: fill beside the step end" SXMC_LDS_RESERVE=20480 SXMC_GATED_STEP=1
done
cat $out
