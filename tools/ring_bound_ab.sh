#!/bin/bash
# One workgroup of 512 lanes per CU (2 waves per SIMD, up to 256 registers each) with a deeper ring of codes, against the
# default shape.  Libraries: make VARIANT=_r8b512 EXTRA="-DSXMC_RING=8 -DSXMC_ORDERED_BOUND=512" (measurement builds).
out=${1:-gpurun_out/ring_bound_ab.log}
run() {
  label=$1; lib=$2; shift; shift
  SXMC_HIP_LIB=$lib python bench.py --also none --experiments 0 --steps 300 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-28s %8.1f evals/s  fill %.1f us  step %.1f us  %s' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step'], r['config']['launch_plan'][-22:]))" >> $out
}
: > $out
D=sxmc_amd/csrc
for k in 1 2; do
  run "default" $D/libsxmc_hip.so --no-autotune
  run "ring 4, bound 512, 512x1" $D/libsxmc_hip_r4b512.so --no-autotune --launch 512,1
  run "ring 8, bound 512, 512x1" $D/libsxmc_hip_r8b512.so --no-autotune --launch 512,1
  run "ring 16, bound 512, 512x1" $D/libsxmc_hip_r16b512.so --no-autotune --launch 512,1
  run "ring 8, bound 512, 256x2" $D/libsxmc_hip_r8b512.so --no-autotune --launch 256,2
done
cat $out
