#!/bin/bash
# More measurement hooks of the fill over codes (config 3).  usage: tools/codes_bounds2.sh [out-file]
out=${1:-gpurun_out/codes_bounds2.log}
run() {
  label=$1; shift
  python bench.py --also none --experiments 0 --steps 200 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-34s %8.1f evals/s  fill %.1f us  step %.1f us  lanes %s' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step'], r['config'].get('autotuned_lanes_per_cu')))" >> $out
}
: > $out
run "codes 1024x1" --launch 1024,1 --no-autotune
run "codes 1024x1 no drain (16)" --launch 1024,1 --no-autotune --debug-mode 16
run "codes 1024x1 stream-only (1)" --launch 1024,1 --no-autotune --debug-mode 1
run "codes 1024x1 no hist, no drain (20)" --launch 1024,1 --no-autotune --debug-mode 20
run "codes 768x1" --launch 768,1 --no-autotune
run "codes 768x1 no drain (16)" --launch 768,1 --no-autotune --debug-mode 16
run "codes 896x1" --launch 896,1 --no-autotune
cat $out
