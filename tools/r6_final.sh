#!/bin/bash
# Round-6 profile bundle (GPU box): C4 default line + trace + PMC passes; bench + trace for the other configs; text ingest at 1e9
# lines (one sort pipeline) and at 5e9 lines (C5's input, user-id range by user-id range) with the partitioned finish's kernel
# trace at 2.5e9 records; top-N at 1M items, bulk calls and request threads, with its kernel trace.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export TMPDIR=/tmp
tools/profile_round.sh r6c4 > gpurun_out/r6c4.log 2>&1
DB=$(find gpurun_out/r6c4/trace -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > gpurun_out/r6c4/bench_kernel_stats.txt 2>&1
rm -rf gpurun_out/r6c4/trace
python tools/pmc_to_json.py gpurun_out/r6c4 gpurun_out/r6c4/r6_c4 c4 64 > gpurun_out/r6c4/pmc_to_json.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/r6c4/pmc_traffic.json 2>/dev/null
find gpurun_out/r6c4 -name "*.csv" -delete
for wl in c5rank c2 c3 k30 c4rank; do tools/profile_workload.sh r6w $wl > /dev/null 2>&1; done
mkdir -p gpurun_out/r6x
python tools/bench_ingest.py --from-text --records 1000000000 --repeat 2 > gpurun_out/r6x/r6_ingest_text_1e9_bench.json 2> gpurun_out/r6x/ingest.err
python tools/bench_ingest.py --records 1000000000 > gpurun_out/r6x/r6_ingest_1e9_bench.json 2>> gpurun_out/r6x/ingest.err
python tools/bench_ingest.py --from-text --stream-text --records 5000000000 --users 100000000 --items 10000000 > gpurun_out/r6x/r6_ingest_text_5e9_bench.json 2>> gpurun_out/r6x/ingest.err
(cd /tmp; rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r6x/big_trace -- python $ROOT/tools/ingest_big_probe.py > $ROOT/gpurun_out/r6x/r6_ingest_big_2p5e9_probe.json 2>/dev/null)
python tools/rocprof_summary.py $(find gpurun_out/r6x/big_trace -name "*.db" | head -1) --top 45 > gpurun_out/r6x/r6_ingest_big_2p5e9_kernel_stats.txt 2>&1
rm -rf gpurun_out/r6x/big_trace
python tools/bench_topn.py > gpurun_out/r6x/r6_topn_1M_bench.json 2> gpurun_out/r6x/topn.err
(cd /tmp; rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r6x/topn_trace -- python $ROOT/tools/bench_topn.py --no-cpu-baseline > /dev/null 2>&1)
python tools/rocprof_summary.py $(find gpurun_out/r6x/topn_trace -name "*.db" | head -1) --top 25 > gpurun_out/r6x/r6_topn_1M_kernel_stats.txt 2>&1
rm -rf gpurun_out/r6x/topn_trace
ls -la gpurun_out/r6c4 gpurun_out/r6w gpurun_out/r6x | head -70
