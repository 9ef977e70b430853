#!/bin/bash
# Deeper rings now that the stream loops' waits are the counted ones (fill_kernels.inc.h, SXMC_BOX_RING / SXMC_ORD_RING):
# boxed 8 -> 16 units of 8 bytes; ordered 4 -> 8 units of 16 bytes with a launch bound of 768 lanes (145 VGPRs).
# Libraries: make -C sxmc_amd/csrc VARIANT=_bx_r16 EXTRA=-DSXMC_BOX_RING=16;
#            make -C sxmc_amd/csrc VARIANT=_ord_r8 EXTRA="-DSXMC_ORD_RING=8 -DSXMC_ORDERED_BOUND=768".  One box, alternating.
out=${1:-gpurun_out/ring_depth_ab.log}
run() {
  label=$1; lib=$2; shift; shift
  SXMC_HIP_LIB=$lib python bench.py --also none --experiments 0 --steps 300 --no-cpu-baseline --no-autotune "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-40s %8.1f evals/s  fill %.2f us  step %.2f us  %s' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step'], r['config']['launch_plan'][-44:]))" >> $out
}
: > $out
D=sxmc_amd/csrc
for k in 1 2 3; do
  run "boxed, ring 8 (product)" $D/libsxmc_hip.so
  run "boxed, ring 16" $D/libsxmc_hip_bx_r16.so
  run "ordered 768x1, ring 4 (product)" $D/libsxmc_hip.so --no-boxes --launch 768,1
  run "ordered 768x1, ring 8" $D/libsxmc_hip_ord_r8.so --no-boxes --launch 768,1
  run "ordered default shape, ring 4" $D/libsxmc_hip.so --no-boxes
done
cat $out
