#!/bin/bash
# refresh of the C4 part of the round-4 bundle with the final bench.py (default line, traced line, PMC passes)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
tools/profile_round.sh r4c4 > gpurun_out/r4c4.log 2>&1
DB=$(find gpurun_out/r4c4/trace -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > gpurun_out/r4c4/bench_kernel_stats.txt 2>&1
rm -rf gpurun_out/r4c4/trace
python tools/pmc_to_json.py gpurun_out/r4c4 gpurun_out/r4c4/r4_c4 c4 64 > gpurun_out/r4c4/pmc_to_json.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/r4c4/pmc_traffic.json 2>/dev/null
find gpurun_out/r4c4 -name "*.csv" -delete
tail -c 700 gpurun_out/r4c4/bench.json
