#!/bin/bash
# More launch shapes of the fill over codes (workgroups per CU beyond what is resident at once).  usage: tools/codes_shapes2.sh [out]
out=${1:-gpurun_out/codes_shapes2.log}
run() {
  label=$1; shift
  python bench.py --also none --steps 300 --no-cpu-baseline --experiments 8 --exp-lockstep 0 "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']; e=r.get('experiments') or {}
print('%-22s %8.1f evals/s  fill %.1f us  %s | 8 in flight: %s steps/s' % ('$label', r['value'], 1e3*f['avg_launch_ms'], r['config']['launch_plan'][-24:], e.get('steps_per_sec_inside')))" >> $out
}
: > $out
run "512 x 2 (default)" --no-autotune
run "512 x 4" --launch 512,4 --no-autotune
run "256 x 4" --launch 256,4 --no-autotune
run "384 x 2" --launch 384,2 --no-autotune
run "512 x 3" --launch 512,3 --no-autotune
run "512 x 2 (default)" --no-autotune
cat $out
