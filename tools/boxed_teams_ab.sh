#!/bin/bash
# Teams of workgroups over contiguous parts of the sorted table (sxplan::interleaved_segments, SXMC_PART_GROUPS on the
# measurement build) for the boxed fill: a workgroup of a team sees a fraction of the histogram's bins and flushes as many.
out=${1:-gpurun_out/boxed_teams_ab.log}
run() {
  label=$1; n=$2
  SXMC_PART_GROUPS=$n SXMC_HIP_LIB=sxmc_amd/csrc/libsxmc_hip_measure.so python bench.py --also none --experiments 0 --steps 300 --no-cpu-baseline --no-autotune 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-14s %8.1f evals/s  fill %.2f us  step %.2f us  %s' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step'], r['config']['launch_plan'][-20:]))" >> $out
}
: > $out
for k in 1 2; do for n in 1 2 3 5 7 10 20; do run "teams $n" $n; done; done
cat $out
