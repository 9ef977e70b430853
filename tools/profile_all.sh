#!/bin/bash
# Runs ON the GPU box: the rocprofv3 evidence of a round for every workload whose fill bench.py reports -- kernel
# trace + three PMC passes each (tools/profile_on_gpu.sh), summaries under gpurun_out/<prefix>_<workload>/.
# Usage: tools/profile_all.sh <prefix> [c5]   (c5: only config 5, in a call of its own -- a gpurun call is 20 minutes at most)
#             (then copy the summaries into profiles/ and run tools/update_traffic.py)
set -o pipefail
P=$1
if [ "$2" = "c5" ]; then
  bash tools/profile_on_gpu.sh ${P}_c5 --workload c5 --steps 40 --also none --experiments 0 || exit 6
  exit 0
fi
bash tools/profile_on_gpu.sh ${P}_c3_boxed --also none --experiments 0 || exit 8     # (the walk's choice at its first steps: the boxed form)
bash tools/profile_on_gpu.sh ${P}_c3 --no-boxes --also none --experiments 0 || exit 1   # (the ordered form over codes)
bash tools/profile_on_gpu.sh ${P}_c3_no_codes --no-codes --also none --experiments 0 || exit 7
bash tools/profile_on_gpu.sh ${P}_c3_lookahead --lookahead --also none --experiments 0 || exit 2
bash tools/profile_on_gpu.sh ${P}_c3_no_order --no-order --also none --experiments 0 || exit 3
bash tools/profile_on_gpu.sh ${P}_c2 --workload c2 --also none --experiments 0 || exit 4
bash tools/profile_on_gpu.sh ${P}_c2_rows --workload c2 --no-prebin --also none --experiments 0 || exit 5
