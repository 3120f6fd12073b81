#!/bin/bash
# Round profile bundle (run on the GPU box through gpurun): default bench line, rocprofv3
# kernel-trace stats of the SAME command, and separate PMC passes for HBM traffic.
# usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r1}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py > $OUT/bench_traced.json 2> $OUT/trace.err
# counters in their own runs, no tracing domains (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-unplanted --no-fp32-leg > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_64B --output-format csv -d $OUT/pmc_write -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-unplanted --no-fp32-leg > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq0 -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-unplanted --no-fp32-leg > /dev/null 2> $OUT/pmc_sq0.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq1 -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-unplanted --no-fp32-leg > /dev/null 2> $OUT/pmc_sq1.err
find $OUT -name "*.csv" -size +20M -delete   # keep the merge-back under the 64 MiB cap
ls -la $OUT $OUT/*/ | head -40
