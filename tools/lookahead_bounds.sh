#!/bin/bash
# Runs ON the GPU box: what bounds the look-ahead pass (two evaluations per pass over the tables)?  The pass as it is,
# then with the measurement hooks of sxmc_group_set_debug_mode (wrong results, right timings): 4 = no histogram update
# (everything but the LDS atomics: the most ANY change to the LDS histograms -- replicas, packed counters -- can gain),
# 2 = no HBM stream (arithmetic + LDS alone), 1 = stream alone.  Same box, one after the other.
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
for dbg in 0 4 2 1 0; do
  f=$OUT/la_dbg${dbg}_$RANDOM.json
  timeout -k 10 300 python3 bench.py --lookahead --steps 600 --warmup 50 --also none --experiments 0 --no-cpu-baseline --debug-mode $dbg > $f 2> $f.err || { echo "dbg $dbg failed"; tail -3 $f.err; exit 1; }
  python3 - $f $dbg <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf = r["roofline"]
print("debug %s: pass %.1f us (%d launches)  %.0f steps/s  %s" % (sys.argv[2], 1e3 * rf["avg_launch_ms"], rf["launches_timed"], r["value"], (r["config"].get("lookahead") or {}).get("steps_per_pass")))
PY
done
