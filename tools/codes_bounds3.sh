#!/bin/bash
# The LDS additions of the fill over codes (config 3): with, without (hook 32), all into one word per granule (hook 4).
out=${1:-gpurun_out/codes_bounds3.log}
run() {
  label=$1; shift
  python bench.py --also none --experiments 0 --steps 200 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-40s %8.1f evals/s  fill %.1f us  step %.1f us' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step']))" >> $out
}
: > $out
run "codes 768x1" --launch 768,1 --no-autotune
run "codes 768x1 no LDS additions (32)" --launch 768,1 --no-autotune --debug-mode 32
run "codes 768x1 no LDS additions, no drain (48)" --launch 768,1 --no-autotune --debug-mode 48
run "codes 768x1 no drain (16)" --launch 768,1 --no-autotune --debug-mode 16
run "codes 768x1 stream only (1)" --launch 768,1 --no-autotune --debug-mode 1
run "codes 1024x1" --launch 1024,1 --no-autotune
run "codes 1024x1 no LDS additions (32)" --launch 1024,1 --no-autotune --debug-mode 32
cat $out
