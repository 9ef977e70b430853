#!/bin/bash
# Strata of x - t inside a bucket of a boxed table (SXMC_BOX_STRATA, measurement build), config 3, one box.
out=${1:-gpurun_out/boxed_strata.log}
run() {
  label=$1; n=$2; shift; shift
  SXMC_BOX_STRATA=$n SXMC_HIP_LIB=sxmc_amd/csrc/libsxmc_hip_measure.so python bench.py --also none --experiments 0 --steps 300 --no-cpu-baseline --no-autotune "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-28s %8.1f evals/s  fill %.2f us  step %.2f us' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step']))" >> $out
}
: > $out
for k in 1 2; do
  for n in 1 2 3 4 6 8; do run "strata $n" $n; done
  for n in 1 2 4; do run "strata $n, no drain" $n --debug-mode 16; done
done
cat $out
