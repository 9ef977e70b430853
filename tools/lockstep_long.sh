#!/bin/bash
# steady state of the lockstep sets: experiments long enough that their set-up no longer shows.
# usage: tools/lockstep_long.sh <tag>
set -e
tag=${1:-lslong}
out=gpurun_out/$tag
mkdir -p $out
for cfg in "2 4" "2 2" "3 2" "4 2" "4 1" "2 1"; do
  set -- $cfg
  python3 bench.py --steps 20 --warmup 5 --also none --no-cpu-baseline --exp-steps 20000 --exp-lockstep $1 --exp-sets $2 \
    --experiments 8 > $out/l$1_s$2.json 2> $out/l$1_s$2.err
  python3 - $out/l$1_s$2.json $1 $2 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d["experiments"]
print("chains %s x sets %s: lockstep %.0f steps/s (%.2f s) | separate fills %.0f steps/s" % (
    sys.argv[2], sys.argv[3], e["lockstep"]["steps_per_sec_inside"], e["lockstep"]["seconds"],
    e["separate_fills"]["steps_per_sec_inside"]))
PY
done
