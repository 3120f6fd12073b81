#!/bin/bash
# usage: tools/pmc_run.sh <tag> <bench args...>   -- collects PMC passes (each its own rocprofv3 run,
# counters only, no tracing domains) into gpurun_out/pmc_<tag>/passN
set -u
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"
 "FETCH_SIZE TCC_HIT"
 "WRITE_SIZE TCC_MISS TCC_REQ"
 "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TA_BUSY_avr SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -o p -- python $ROOT/bench.py "$@" > $OUT/pass$i.out 2> $OUT/pass$i.err
  echo "pass $i rc=$?"; tail -2 $OUT/pass$i.err
  i=$((i+1))
done
ls -la $OUT/*/
