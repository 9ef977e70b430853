#!/bin/bash
# Runs ON the GPU box: the three forms of a step, alternating runs on the same box, BASELINE configs 3 and 2 --
#   fused: the whole step ONE launch (fill_step_kernel: the fill's workgroups + the step end's roles in one grid; default)
#   coop : the fill, then the cooperative step end (step_end_kernel)                       SXMC_FUSED_STEP=0
#   two  : the fill, look-ups + event sum, step end + clearing (round 3's three launches)  ... SXMC_COOP_STEP_END=0
# then the device timeline of each form under rocprofv3 --kernel-trace.  Usage: tools/step_forms_ab.sh <tag>
set -o pipefail
TAG=${1:-step_forms_ab}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
val() { python3 -c 'import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.1f evals/s  %.2f us/step  fill alone %.2f us  whole-step frac %.4f  launches/step %s" % (r["value"], 1e3*r["ms_per_step"], 1e3*r["roofline"]["avg_launch_ms"], r["roofline"]["whole_step_frac"], r["config"]["launches_per_step"]))'; }
env_of() { case $1 in fused) echo "SXMC_FUSED_STEP=1";; coop) echo "SXMC_FUSED_STEP=0";; two) echo "SXMC_FUSED_STEP=0 SXMC_COOP_STEP_END=0";; esac; }
for wl in c3 c2; do
  for rep in 1 2 3; do
    for form in fused coop two; do
      echo "$wl $form: $(env $(env_of $form) python3 bench.py --workload $wl --steps 2000 --warmup 100 --also none --experiments 0 --no-cpu-baseline 2>>$OUT/err.log | val)" | tee -a $OUT/ab.log
    done
  done
done
for wl in c3 c2; do
  for form in fused coop; do
    case $form in fused) export SXMC_FUSED_STEP=1;; coop) export SXMC_FUSED_STEP=0;; esac
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_${wl}_$form -- python3 bench.py --workload $wl --steps 400 --warmup 50 --also none --experiments 0 --no-cpu-baseline > $OUT/trace_${wl}_$form.log 2>&1 || exit 2
    echo "== $wl $form" | tee -a $OUT/timelines.txt
    python3 tools/summarize_rocprof.py timeline $OUT/trace_${wl}_$form $OUT/timeline_${wl}_$form.csv 0.5 | tee -a $OUT/timelines.txt
    rm -rf $OUT/trace_${wl}_$form
  done
done
