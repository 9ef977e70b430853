#!/bin/bash
# Runs ON the GPU box: the driver's command (--steps 20 --warmup 5) with different amounts of untimed pre-warming --
# how long does the device take to reach its settled clocks under this load?  Usage: tools/prewarm_sweep.sh
val() { python3 -c 'import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.1f evals/s  fill %.2f us (post-timed, %d launches)  in-region fill n/a" % (r["value"], 1e3*r["roofline"]["avg_launch_ms"], r["roofline"]["launches_timed"]))'; }
for rep in 1 2; do
  for pw in 300 1000 2000 4000 8000; do
    echo "prewarm $pw: $(python3 bench.py --steps 20 --warmup 5 --prewarm $pw --no-cpu-baseline --experiments 0 --also none 2>/dev/null | val)"
  done
done
