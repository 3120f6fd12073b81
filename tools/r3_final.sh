#!/bin/bash
# Round-3 profile bundle (GPU box): C4 default line + trace + PMC passes; bench + trace for the other configs;
# per-dispatch HBM traffic of the C5 rank at its true shape.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
tools/profile_round.sh r3c4 > gpurun_out/r3c4.log 2>&1
DB=$(find gpurun_out/r3c4/trace -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > gpurun_out/r3c4/bench_kernel_stats.txt 2>&1
rm -rf gpurun_out/r3c4/trace
python tools/pmc_to_json.py gpurun_out/r3c4 gpurun_out/r3c4/r3_c4 c4 64 > gpurun_out/r3c4/pmc_to_json.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/r3c4/pmc_traffic.json 2>/dev/null
find gpurun_out/r3c4 -name "*.csv" -delete
for wl in c5shard8 c5rank c2 c3 k30; do tools/profile_workload.sh r3w $wl > /dev/null 2>&1; done
tools/pmc_traffic_quick.sh c5rank --workload c5rank > /dev/null 2>&1
tools/pmc_traffic_quick.sh c5shard8 --workload c5shard8 > /dev/null 2>&1
ls -la gpurun_out/r3c4 gpurun_out/r3w | head -60
