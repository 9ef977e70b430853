#!/bin/bash
# Runs ON the GPU box: where do kernel arguments live?  HIP_FORCE_DEV_KERNARG=1 (device memory) is the default of this
# ROCm (7.2): unset and 1 give the same config-2 step (21.6 us), 0 (host memory) costs the fill 1 us (10.8 against 9.8).
# Nothing to set; kept as the record of the check (round 3).
set -o pipefail
mkdir -p gpurun_out/r3k2
for v in default 1 0 default 1 0; do
  f=gpurun_out/r3k2/ka_${v}_$RANDOM.json
  if [ $v = default ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  timeout -k 10 300 python3 bench.py --workload c2 --steps 2000 --warmup 100 --also none --experiments 0 --no-cpu-baseline > $f 2> $f.err || { echo failed; tail -3 $f.err; exit 1; }
  python3 - $f $v <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf = r["roofline"]
print("HIP_FORCE_DEV_KERNARG=%-7s C2: %.0f evals/s  step %.2f us  fill %.2f us" % (sys.argv[2], r["value"], 1e3 * r["ms_per_step"], 1e3 * rf["avg_launch_ms"]))
PY
done
