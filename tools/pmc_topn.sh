#!/bin/bash
# two SQ counter passes over one 4096-query top-N call (one slot: the kernels one after the other), summarised per kernel
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_topn
rm -rf $OUT; mkdir -p $OUT
export MALS_TOPN_SLOTS=1
test -f /tmp/tn_one.py || (bash $ROOT/tools/prof_topn.sh > /dev/null 2>&1)
cd /tmp
ROOT=$ROOT rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pass0 -o p -- python /tmp/tn_one.py > $OUT/pass0.out 2> $OUT/pass0.err
ROOT=$ROOT rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pass1 -o p -- python /tmp/tn_one.py > $OUT/pass1.out 2> $OUT/pass1.err
python $ROOT/tools/pmc_summary.py $OUT "topn_stream_kernel<2, 4, 1" > $OUT/summary.txt
find $OUT -name "*.csv" -delete
