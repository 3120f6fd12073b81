#!/bin/bash
# usage: tools/pmc_run2.sh <tag> "<counters>" <bench args...>  -- one PMC pass
set -u
TAG=$1; shift
CNT=$1; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT/pass0 -o p -- python $ROOT/bench.py "$@" > $OUT/pass0.out 2> $OUT/pass0.err
echo "rc=$?"; tail -2 $OUT/pass0.err
