#!/bin/bash
# Runs ON the GPU box: the headline (config 3) from the round-2 tree and from this tree, alternating on the same box:
# is the fill as fast as it was?  The old tree is exported and built first, in the container (build/ is git-ignored
# but travels with gpurun):
#   mkdir -p build/r02tree && git archive 6ca6f10 | tar -x -C build/r02tree && make -C build/r02tree/sxmc_amd/csrc
#   make -C build/r02tree/oracle
# Result of round 3 (gpurun_out/r3g): old 125.1 / 132.2 us, new 131.9 / 130.4 us -- unchanged within the run-to-run spread.
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
ARGS="--steps 400 --warmup 50 --also none --experiments 0 --no-cpu-baseline"
for i in 1 2; do
  (cd build/r02tree && timeout -k 10 300 python3 bench.py $ARGS > $OUT/old_$i.json 2> $OUT/old_$i.err) || { echo old failed; tail -3 $OUT/old_$i.err; exit 1; }
  timeout -k 10 300 python3 bench.py $ARGS > $OUT/new_$i.json 2> $OUT/new_$i.err || { echo new failed; tail -3 $OUT/new_$i.err; exit 1; }
done
python3 - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    r = json.loads(open(f).read().strip().splitlines()[-1]); rf = r["roofline"]
    print("%-10s %7.0f evals/s  step %.1f us  fill %.1f us (%d launches)  frac %.3f" % (f.split("/")[-1], r["value"], 1e3 * r["ms_per_step"], 1e3 * rf["avg_launch_ms"], rf["launches_timed"], rf["frac"]))
PY
