#!/bin/bash
# Fake experiments with a fill per chain (8 in flight): do two chains' fills share a CU when each leaves LDS for the other?
out=${1:-gpurun_out/codes_concurrent.log}
run() {
  label=$1; shift
  python bench.py --also none --steps 100 --no-cpu-baseline --experiments 8 --exp-lockstep 0 "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']; e=r.get('experiments') or {}
print('%-44s headline %8.1f /s fill %.1f us | experiments %.2f /s  %s steps/s inside' % ('$label', r['value'], 1e3*f['avg_launch_ms'], e.get('experiments_per_sec', 0), e.get('steps_per_sec_inside')))" >> $out
}
: > $out
run "4 replicas, autotuned"
SXMC_ORDERED_REPLICAS_LOG2=1 run "2 replicas, 512 x 1" --launch 512,1 --no-autotune
SXMC_ORDERED_REPLICAS_LOG2=1 run "2 replicas, 512 x 2" --launch 512,2 --no-autotune
SXMC_ORDERED_REPLICAS_LOG2=1 run "2 replicas, 768 x 1" --launch 768,1 --no-autotune
SXMC_ORDERED_REPLICAS_LOG2=0 run "1 replica, 512 x 1" --launch 512,1 --no-autotune
SXMC_ORDERED_REPLICAS_LOG2=0 run "1 replica, 256 x 4" --launch 256,4 --no-autotune
cat $out
