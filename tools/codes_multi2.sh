#!/bin/bash
# The look-ahead pass (two chains per pass) over codes with the measurement hooks.  usage: tools/codes_multi2.sh [out]
out=${1:-gpurun_out/codes_multi2.log}
run() {
  label=$1; shift
  python bench.py --also none --steps 300 --no-cpu-baseline --experiments 0 --lookahead "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-36s %8.1f steps/s  pass %.1f us  step %.1f us' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step']))" >> $out
}
: > $out
run "look-ahead over codes"
run "... no LDS additions (32)" --debug-mode 32
run "... no drain (16)" --debug-mode 16
run "... stream only (1)" --debug-mode 1
run "... no LDS additions, no drain (48)" --debug-mode 48
cat $out
