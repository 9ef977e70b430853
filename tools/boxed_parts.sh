#!/bin/bash
# Parts of fill_boxed_kernel at config 3 (measurement build, kernel hooks: RESULTS ARE WRONG with any of them set).
out=${1:-gpurun_out/boxed_parts.log}
run() {
  label=$1; shift
  SXMC_HIP_LIB=sxmc_amd/csrc/libsxmc_hip_measure.so python bench.py --also none --experiments 0 --steps 200 --no-cpu-baseline --no-autotune "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-44s fill %.2f us  step %.2f us' % ('$label', 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step']))" >> $out
}
: > $out
for k in 1 2; do
  run "everything" --debug-mode 0
  run "the stream alone (1)" --debug-mode 1
  run "queues dropped: no drain (16)" --debug-mode 16
  run "no LDS additions (32)" --debug-mode 32
  run "no additions, no drain (48)" --debug-mode 48
  run "stream alone, no drain (17)" --debug-mode 17
done
cat $out
