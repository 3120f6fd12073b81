#!/bin/bash
# Round-4 profile bundle (GPU box): C4 default line + trace + PMC passes; bench + trace for the other configs;
# per-dispatch HBM traffic of the C5 rank at its true shape (the LDS-staged k = 128 kernels).
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
for wl in c5rank c2 c3 k30 c4rank; do tools/profile_workload.sh r4w $wl > /dev/null 2>&1; done
tools/pmc_traffic_quick.sh c5rank --workload c5rank > /dev/null 2>&1
ls -la gpurun_out/r4c4 gpurun_out/r4w | head -60
