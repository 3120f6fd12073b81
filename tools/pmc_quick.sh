#!/bin/bash
# usage: tools/pmc_quick.sh <tag> <bench args...>  -- two SQ counter passes, summarised per kernel
set -u
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pass0 -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/pass0.out 2> $OUT/pass0.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pass1 -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/pass1.out 2> $OUT/pass1.err
python $ROOT/tools/pmc_summary.py $OUT
find $OUT -name "*.csv" -size +5M -delete
