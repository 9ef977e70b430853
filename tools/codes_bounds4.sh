#!/bin/bash
# The parts of the fill over codes at its default shape (two workgroups of 512 lanes per CU), config 3, one box:
# whole | no LDS additions (hook 32) | no drain (16) | neither (48) | stream only (1).
out=${1:-gpurun_out/codes_bounds4.log}
run() {
  label=$1; shift
  python bench.py --also none --experiments 0 --steps 200 --no-cpu-baseline --no-autotune "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-44s %8.1f evals/s  fill %.1f us  step %.1f us  %s' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step'], r['config']['launch_plan'][-22:]))" >> $out
}
: > $out
for k in 1 2; do
run "codes (default shape)"
run "codes, no LDS additions (32)" --debug-mode 32
run "codes, no drain (16)" --debug-mode 16
run "codes, no LDS additions, no drain (48)" --debug-mode 48
run "codes, stream only (1)" --debug-mode 1
done
cat $out
