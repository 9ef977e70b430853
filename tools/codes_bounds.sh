#!/bin/bash
# What bounds the fill over codes (config 3): the measurement hooks of sxmc_group_set_debug_mode and launch shapes.
# usage: tools/codes_bounds.sh [out-file]   (on the GPU box; prints evals/s and the fill's mean duration per setting)
out=${1:-gpurun_out/codes_bounds.log}
run() {
  label=$1; shift
  python bench.py --also none --experiments 0 --steps 200 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-34s %8.1f evals/s  fill %.1f us  lanes %s  plan %s' % ('$label', r['value'], 1e3*f['avg_launch_ms'], r['config'].get('autotuned_lanes_per_cu'), r['config']['launch_plan']))" >> $out
}
: > $out
run "codes"
run "codes stream-only (1)" --debug-mode 1
run "codes no-stream (2)" --debug-mode 2
run "codes no-hist-update (4)" --debug-mode 4
run "codes 512x1" --launch 512,1 --no-autotune
run "codes 768x1" --launch 768,1 --no-autotune
run "codes 1024x1" --launch 1024,1 --no-autotune
SXMC_CODES=0 run "floats"
SXMC_CODES_QUEUE_LOG=0 run "codes, no queue"
cat $out
