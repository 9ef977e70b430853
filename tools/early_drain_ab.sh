#!/bin/bash
# The early, overlapped drain of the queues of the fill over codes (fill_ordered_body: kDrainEvery / kDrainFinal), one
# box, alternating.  Libraries: make -C sxmc_amd/csrc VARIANT=_e32 EXTRA=-DSXMC_DRAIN_EVERY=32 etc. (see the list below).
out=${1:-gpurun_out/early_drain_ab.log}
run() {
  label=$1; lib=$2; shift; shift
  SXMC_HIP_LIB=$lib python bench.py --also none --experiments 0 --steps 300 --no-cpu-baseline --no-autotune "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-34s %8.1f evals/s  fill %.2f us  step %.2f us  %s' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step'], r['config']['launch_plan'][-22:]))" >> $out
}
: > $out
D=sxmc_amd/csrc
for k in 1 2 3; do
  run "product (no early drain, final 1)" $D/libsxmc_hip.so
  run "every 64" $D/libsxmc_hip_e64.so
  run "every 32" $D/libsxmc_hip_e32.so
  run "final batch 3" $D/libsxmc_hip_f3.so
  run "every 32 + final batch 3" $D/libsxmc_hip_e32f3.so
  run "every 16 + final batch 2" $D/libsxmc_hip_e16f2.so
done
cat $out
