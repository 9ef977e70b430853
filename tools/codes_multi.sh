#!/bin/bash
# The lockstep / look-ahead passes over codes against the float columns (config 3).  usage: tools/codes_multi.sh [out]
out=${1:-gpurun_out/codes_multi.log}
run() {
  label=$1; shift
  python bench.py --also none --steps 300 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']; e=r.get('experiments') or {}
ls=(e.get('lockstep') or {}); sf=(e.get('separate_fills') or {})
print('%-28s %8.1f /s  fill %.1f us  step %.1f us | lockstep %s steps/s  separate %s steps/s' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step'], ls.get('steps_per_sec_inside'), sf.get('steps_per_sec_inside')))" >> $out
}
: > $out
run "lookahead codes" --lookahead --experiments 0
SXMC_CODES=0 run "lookahead floats" --lookahead --experiments 0
run "ensemble codes" --experiments 8
SXMC_CODES=0 run "ensemble floats" --experiments 8
run "ensemble codes 2x4" --experiments 8 --exp-lockstep 4 --exp-sets 2
cat $out
