#!/bin/bash
# Runs ON the GPU box: does a longer untimed run-in (--prewarm) change the headline?  Round 3, two boxes: on one,
# 4 000 steps of run-in gave 7 050-7 150 evals/s against 6 800-6 880 with the default 300 in three alternating pairs
# (the fill the same, the rest of the step 4 us shorter); on the other no pattern (6 830-7 190 for 300 ... 16 000).
# Within the +-2.5 % run-to-run spread of a box: the default stays 300.   PW_LIST="300 4000 ..." overrides the series.
set -o pipefail
mkdir -p gpurun_out/r3w
for pw in ${PW_LIST:-300 4000 300 4000 300 4000}; do
  f=gpurun_out/r3w/pw${pw}_$RANDOM.json
  timeout -k 10 300 python3 bench.py --steps 400 --warmup 50 --prewarm $pw --also none --experiments 0 --no-cpu-baseline > $f 2> $f.err || { echo failed; tail -3 $f.err; exit 1; }
  python3 - $f $pw <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf = r["roofline"]
print("prewarm %5s: %.0f evals/s  step %.1f us  fill %.2f us (in-region %.2f)  frac %.3f" % (sys.argv[2], r["value"], 1e3 * r["ms_per_step"], 1e3 * rf["avg_launch_ms"], 1e3 * rf["in_timed_region"]["avg_launch_ms"], rf["frac"]))
PY
done
