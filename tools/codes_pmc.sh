#!/bin/bash
# PMC counters of the fill over codes (config 3), one pass.  usage: tools/codes_pmc.sh <tag> [bench args]
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
B="python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --experiments 0 --also none $@"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || exit 4
python3 tools/summarize_rocprof.py pmc $OUT/pmc_summary.csv $OUT/pmc_sq $OUT/pmc_sq2 > /dev/null
rm -rf $OUT/pmc_sq $OUT/pmc_sq2
grep -E "Kernel|fill_" $OUT/pmc_summary.csv
