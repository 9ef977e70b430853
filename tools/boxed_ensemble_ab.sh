#!/bin/bash
# The boxed fill's launch shape with 8 chains in flight (SXMC_BOX_LANES, measurement build): 16 experiments of 2 000 steps
# and the single walk, one box, alternating.
out=${1:-gpurun_out/boxed_ensemble_ab.log}
run() {
  label=$1; lanes=$2
  SXMC_BOX_LANES=$lanes tests/cpp/bench_cpp_measure --walks auto=4000 --experiments 16 --exp-steps 2000 --sets 8 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): continue
    r=json.loads(l)
    if r.get('leg')=='ensemble' or 'experiments_per_sec' in r:
        print('%-18s ensemble %.3f exp/s  %.0f steps/s inside' % ('$label', r['experiments_per_sec'], r['steps_per_sec_inside']))
    elif 'steps_per_sec' in r:
        print('%-18s walk %s %.0f steps/s' % ('$label', r.get('walk'), r['steps_per_sec']))" >> $out
}
: > $out
for k in 1 2; do
  run "1024 x 1" 1024
  run "768 x 1" 768
  run "512 x 2" 512
done
cat $out
