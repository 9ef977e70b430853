#!/bin/bash
# Runs ON the GPU box: BASELINE config 2 (10^7 samples, no systematics) under launch shapes x table forms x column-load
# policies, one bench.py run each on the same box; prints evals/s, the fill's event-timed duration and its roofline
# fraction (on the bytes that form must stream).  Usage: tools/c2_sweep.sh <tag> [units] [extra bench args]
#   units: space-separated list of SXMC_TWO_UNITS values (1 = two units in flight per lane, the product's choice for
#   short launches; 0 = one), default "1 0".  (The cached-loads A/B of profiles/r03_c2_sweep_policy_x_shape.log used a
#   run-time switch that has since been replaced by a build flag: make VARIANT=_cached EXTRA=-DSXMC_CACHED_LOADS=1.)
set -o pipefail
TAG=$1; shift
POLICIES=${1:-"1 0"}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
for form in prebinned rows; do
  extra=""; [ $form = rows ] && extra="--no-prebin"
  for policy in $POLICIES; do
    for launch in 0,0; do
      f=$OUT/c2_${form}_p${policy}_${launch/,/x}.json
      SXMC_TWO_UNITS=$policy timeout -k 10 200 python3 bench.py --workload c2 --also none --experiments 0 --no-cpu-baseline \
        --steps 2000 --warmup 100 --launch $launch $extra "$@" > $f 2> $f.err || { echo "FAILED $form units $policy launch $launch"; tail -3 $f.err; exit 1; }
      python3 - "$f" "$form" "$policy" "$launch" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print("%-9s in-flight %s launch %-7s  %8.0f evals/s  step %.2f us  fill %.2f us  %.1f B/sample  frac %.3f (at 8 B/sample: %.3f)  %s" % (
    sys.argv[2], sys.argv[3], sys.argv[4], r["value"], 1e3 * r["ms_per_step"], 1e3 * rf["avg_launch_ms"],
    rf["bytes_per_sample"], rf["frac"], rf["achieved_at_survey_bytes"] / rf["peak"],
    " ".join(x for x in rf["launch_plan"][0].split() if x.split("=")[0] in ("table", "threads", "grid", "loads"))))
PY
    done
  done
done
