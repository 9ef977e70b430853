#!/bin/bash
# Launch shapes of the fill over codes on ONE box, alternating: 4 replicas x one workgroup of 768 / 1024 per CU against
# 2 replicas x two workgroups of 512.  usage: tools/codes_shapes_ab.sh [out]
out=${1:-gpurun_out/codes_shapes_ab.log}
run() {
  label=$1; shift
  python bench.py --also none --steps 300 --no-cpu-baseline --experiments 8 --exp-lockstep 0 "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']; e=r.get('experiments') or {}
print('%-28s %8.1f evals/s  fill %.1f us  step %.1f us | 8 experiments in flight: %s steps/s' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step'], e.get('steps_per_sec_inside')))" >> $out
}
: > $out
for k in 1 2 3; do
  run "4 replicas, 768 x 1" --launch 768,1 --no-autotune
  SXMC_ORDERED_REPLICAS_LOG2=1 run "2 replicas, 512 x 2" --launch 512,2 --no-autotune
  run "4 replicas, 1024 x 1" --launch 1024,1 --no-autotune
done
cat $out
