#!/bin/bash
# Variants of fill_boxed_kernel (drain batch, mid-stream drains, metadata prefetch), one box, alternating.
# Libraries: make -C sxmc_amd/csrc VARIANT=_bx_d4 EXTRA=-DSXMC_BOX_DRAIN=4 etc. (the list below).
out=${1:-gpurun_out/boxed_ab.log}
run() {
  label=$1; lib=$2; shift; shift
  SXMC_HIP_LIB=$lib python bench.py --also none --experiments 0 --steps 300 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-34s %8.1f evals/s  fill %.2f us  step %.2f us  %s' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step'], r['config']['launch_plan'][-40:]))" >> $out
}
: > $out
D=sxmc_amd/csrc
for k in 1 2; do
  run "product" $D/libsxmc_hip.so
  for v in "$@"; do
    [ "$v" = "$1" ] && continue
    run "$v" $D/libsxmc_hip_bx_$v.so
  done
  run "product, 512x2" $D/libsxmc_hip.so --launch 512,2 --no-autotune
  run "product, 768x1" $D/libsxmc_hip.so --launch 768,1 --no-autotune
  run "product, 1024x1" $D/libsxmc_hip.so --launch 1024,1 --no-autotune
done
cat $out
