#!/bin/bash
# A/B/C of builds of the library on ONE box, alternating.  usage: tools/codes_abc.sh out "lib1 lib2 ..." [bench args]
# ("now" = the product; any other name N = sxmc_amd/csrc/libsxmc_hip_N.so, from `make VARIANT=_N` of the tree to compare)
out=$1; libs=$2; shift; shift
run() {
  label=$1; shift
  python bench.py --also none --experiments 0 --steps 300 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1]); f=r['roofline']
print('%-12s %8.1f evals/s  fill %.1f us  step %.1f us' % ('$label', r['value'], 1e3*f['avg_launch_ms'], 1e3*r['ms_per_step']))" >> $out
}
: > $out
for k in 1 2 3; do
  for l in $libs; do
    if [ "$l" = "now" ]; then run now "$@"; else SXMC_HIP_LIB=sxmc_amd/csrc/libsxmc_hip_$l.so run $l "$@"; fi
  done
done
cat $out
