#!/bin/bash
# Runs ON the GPU box: config 3's headline fill with the member's workgroups split into G teams over contiguous parts of
# the (bin-sorted) bucketed table (sxplan::interleaved_segments, SXMC_PART_GROUPS=G): fewer non-zero bins per workgroup
# -> fewer memory-side atomics in the flush, against coarser interleaving of the stream.  Same box, one after the other.
set -o pipefail
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; shift
for G in ${GROUPS_LIST:-1 2 3 5 7 11 21 1}; do
  f=$OUT/g${G}_$RANDOM.json
  SXMC_PART_GROUPS=$G timeout -k 10 300 python3 bench.py --steps 400 --warmup 50 --also none --experiments 0 --no-cpu-baseline "$@" > $f 2> $f.err || { echo "G=$G failed"; tail -3 $f.err; exit 1; }
  python3 - $f $G <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf = r["roofline"]
print("teams %2s: %.0f evals/s  step %.1f us  fill %.2f us  frac %.3f" % (sys.argv[2], r["value"], 1e3 * r["ms_per_step"], 1e3 * rf["avg_launch_ms"], rf["frac"]))
PY
done
