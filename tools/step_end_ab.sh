#!/bin/bash
# Runs ON the GPU box: the step end as ONE cooperative launch (step_end_kernel, the default) against the two-launch
# form (SXMC_COOP_STEP_END=0), alternating runs on the same box, BASELINE configs 3 and 2; then the device timeline of
# each form under rocprofv3 --kernel-trace (kernel durations + idle gaps).  Usage: tools/step_end_ab.sh <tag>
set -o pipefail
TAG=${1:-step_end_ab}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
val() { python3 -c 'import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.1f evals/s  %.2f us/step  fill %.2f us  whole-step frac %.4f  launches/step %s" % (r["value"], 1e3*r["ms_per_step"], 1e3*r["roofline"]["avg_launch_ms"], r["roofline"]["whole_step_frac"], r["config"]["launches_per_step"]))'; }
for wl in c3 c2; do
  for rep in 1 2 3; do
    for coop in 1 0; do
      echo "$wl coop=$coop: $(SXMC_COOP_STEP_END=$coop python3 bench.py --workload $wl --steps 2000 --warmup 100 --also none --experiments 0 --no-cpu-baseline 2>>$OUT/err.log | val)" | tee -a $OUT/ab.log
    done
  done
done
for wl in c3 c2; do
  for coop in 1 0; do
    SXMC_COOP_STEP_END=$coop timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_${wl}_$coop -- python3 bench.py --workload $wl --steps 400 --warmup 50 --also none --experiments 0 --no-cpu-baseline > $OUT/trace_${wl}_$coop.log 2>&1 || exit 2
    echo "== $wl coop=$coop" | tee -a $OUT/timelines.txt
    python3 tools/summarize_rocprof.py timeline $OUT/trace_${wl}_$coop $OUT/timeline_${wl}_coop$coop.csv 0.5 | tee -a $OUT/timelines.txt
    rm -rf $OUT/trace_${wl}_$coop
  done
done
