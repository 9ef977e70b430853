#!/bin/bash
# Teams of workgroups (SXMC_PART_GROUPS, measurement build) for the ORDERED form over codes at config 3.
out=${1:-gpurun_out/ordered_teams_ab.log}
run() {
  label=$1; n=$2; shift; shift
  SXMC_PART_GROUPS=$n SXMC_HIP_LIB=sxmc_amd/csrc/libsxmc_hip_measure.so python bench.py --also none --experiments 0 --steps 300 --no-cpu-baseline --no-autotune --no-boxes "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-22s %8.1f evals/s  fill %.2f us  step %.2f us  %s' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step'], r['config']['launch_plan'][-32:]))" >> $out
}
: > $out
for k in 1 2; do
  for n in 1 3 7 20; do run "512x2, teams $n" $n; done
  for n in 1 20; do run "768x1, teams $n" $n --launch 768,1; done
done
cat $out
