#!/bin/bash
# Runs ON the GPU box: BASELINE config 2 (80 MB table, no systematics) under launch shapes x column-load policies, one
# bench.py run each on the same box; prints evals/s, the fill's event-timed duration and its roofline fraction.
# Usage: tools/c2_sweep.sh <tag> [extra bench args]
set -o pipefail
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
for policy in 0 1 2; do
  for launch in 0,0 1024,1 768,1 512,1 256,4 1024,2; do
    f=$OUT/c2_p${policy}_${launch/,/x}.json
    SXMC_LOAD_POLICY=$policy timeout -k 10 200 python3 bench.py --workload c2 --also none --experiments 0 --no-cpu-baseline \
      --steps 2000 --warmup 100 --launch $launch "$@" > $f 2> $f.err || { echo "FAILED policy $policy launch $launch"; tail -3 $f.err; exit 1; }
    python3 - "$f" "$policy" "$launch" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print("policy %s launch %-7s  %8.0f evals/s  step %.2f us  fill %.2f us (in-region %.2f)  frac %.3f  %s" % (
    sys.argv[2], sys.argv[3], r["value"], 1e3 * r["ms_per_step"], 1e3 * rf["avg_launch_ms"],
    1e3 * rf["in_timed_region"]["avg_launch_ms"], rf["frac"], rf["launch_plan"][0][:90]))
PY
  done
done
