#!/bin/bash
out=gpurun_out/boxed_replicas_ab.log
run() {
  label=$1; r=$2
  SXMC_ORDERED_REPLICAS_LOG2=$r SXMC_HIP_LIB=sxmc_amd/csrc/libsxmc_hip_measure.so python bench.py --also none --experiments 0 --steps 300 --no-cpu-baseline --no-autotune 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-24s %8.1f evals/s  fill %.2f us  step %.2f us' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step']))" >> $out
}
: > $out
for k in 1 2; do run "4 replicas" 2; run "2 replicas" 1; run "1 replica" 0; done
cat $out
