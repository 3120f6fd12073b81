#!/bin/bash
# C5 rank at its true shape: parity test, bench line, kernel stats, HBM traffic passes (run through gpurun)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3_c5rank
mkdir -p $OUT
cd $ROOT
if [ "${1:-all}" != "prof" ]; then
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "c5_rank" > $OUT/test.log 2>&1; echo "test rc=$?"; tail -5 $OUT/test.log
fi
timeout 600 python bench.py --workload c5rank --steps 5 --warmup 2 > $OUT/c5rank_bench.json 2> $OUT/c5rank_bench.err; echo "bench rc=$?"
tail -c 1500 $OUT/c5rank_bench.json; tail -3 $OUT/c5rank_bench.err
if [ "${1:-all}" != "test" ]; then
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py --workload c5rank --steps 5 --warmup 2 > $OUT/c5rank_bench_traced.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python $ROOT/tools/rocprof_summary.py $DB > $OUT/c5rank_kernel_stats.txt 2>&1 || true
rm -rf $OUT/trace
head -30 $OUT/c5rank_kernel_stats.txt
bash $ROOT/tools/pmc_traffic_quick.sh c5rank --workload c5rank
cat $ROOT/gpurun_out/pmct_c5rank/summary.txt | head -40
fi
