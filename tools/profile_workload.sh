#!/bin/bash
# Bench line + rocprofv3 kernel-trace stats of the same command for one bench workload
# (run on the GPU box through gpurun).  usage: tools/profile_workload.sh <tag> <workload> [bench args]
set -u
TAG=$1; WL=$2; shift 2
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py --workload $WL "$@" > $OUT/${WL}_bench.json 2> $OUT/${WL}_bench.err
tail -c 300 $OUT/${WL}_bench.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${WL}_trace -o bench -- python $ROOT/bench.py --workload $WL --no-cpu-baseline "$@" > $OUT/${WL}_bench_traced.json 2> $OUT/${WL}_trace.err
DB=$(find $OUT/${WL}_trace -name "*.db" | head -1)
python $ROOT/tools/rocprof_summary.py $DB > $OUT/${WL}_kernel_stats.txt 2>&1 || true
rm -rf $OUT/${WL}_trace   # the rocpd database is far above the 64 MiB merge-back cap
head -12 $OUT/${WL}_kernel_stats.txt
