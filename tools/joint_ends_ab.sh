#!/bin/bash
# A/B of the lockstep sets' step ends: two launches for the set (default) against two per chain
# (SXMC_JOINT_STEP_END=0), whole bench.py runs alternated on one box.  usage: tools/joint_ends_ab.sh <tag>
set -e
tag=${1:-jointab}
out=gpurun_out/$tag
mkdir -p $out
for i in 1 2 3; do
  for k in 1 0; do
    SXMC_JOINT_STEP_END=$k python3 bench.py --steps 20 --warmup 5 > $out/joint${k}_$i.json 2> $out/joint${k}_$i.err
    python3 - $out/joint${k}_$i.json $k <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d["experiments"]
c = d["also"]["cpp_host"]["ensemble_lockstep"]
print("joint=%s  python 2 chains x 4 sets: %.0f steps/s (%.3f s) | C++ 4 chains x 2 sets: %.0f steps/s | separate fills %.0f | headline %.0f" % (
    sys.argv[2], e["lockstep"]["steps_per_sec_inside"], e["lockstep"]["seconds"], c["steps_per_sec_inside"],
    e["separate_fills"]["steps_per_sec_inside"], d["value"]))
PY
  done
done
