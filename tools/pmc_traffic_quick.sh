#!/bin/bash
# usage: tools/pmc_traffic_quick.sh <tag> <bench args...>  -- HBM read/write bytes per kernel (two TCC passes)
set -u
TAG=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmct_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pass0 -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/pass0.out 2> $OUT/pass0.err
rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_64B --output-format csv -d $OUT/pass1 -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/pass1.out 2> $OUT/pass1.err
python $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt
grep '^{' $OUT/pass0.out | tail -1 > $OUT/bench_line.json
find $OUT -name "*.csv" -delete
rm -rf $OUT/pass0 $OUT/pass1
