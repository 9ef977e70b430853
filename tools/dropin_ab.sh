#!/bin/bash
# The unchanged caller's walk (mcmc.cpp:264-271 + 314-348 as written) at config 3, A/B over the library's switches for
# it; bench_cpp_measure = the same driver over the measurement build (whose A/B environment switches are live).
out=${1:-gpurun_out/dropin_ab.log}
: > $out
run() {
  label=$1; shift
  env "$@" tests/cpp/bench_cpp_measure --walks reference=3000 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        r=json.loads(ln)
        if r.get('walk')=='reference': print('%-44s %8.1f steps/s  (stepping alone %8.1f)  deferred launches %d' % ('$label', r['steps_per_sec'], r['steps_per_sec_stepping'], r['deferred_launches']))" >> $out
}
for k in 1 2; do
  run "default (batch graph, lazy finish)" SXMC_X=1
  run "no batch graph" SXMC_BATCH_GRAPH=0
  run "no lazy finish" SXMC_LAZY_FINISH=0
  run "no batch graph, no lazy finish" SXMC_BATCH_GRAPH=0 SXMC_LAZY_FINISH=0
done
cat $out
