#!/bin/bash
# Round-2 profile bundle (GPU box): C4 default line + trace + PMC passes; bench + trace for the other configs.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
tools/profile_round.sh r2c4 > gpurun_out/r2c4.log 2>&1
DB=$(find gpurun_out/r2c4/trace -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > gpurun_out/r2c4/bench_kernel_stats.txt 2>&1
rm -rf gpurun_out/r2c4/trace
for wl in c5shard8 c2 c3 k30; do tools/profile_workload.sh r2w $wl > /dev/null 2>&1; done
tools/pmc_quick.sh c5 --workload c5shard8 > gpurun_out/r2_c5shard8_pmc.txt 2>&1
ls -la gpurun_out/r2c4 gpurun_out/r2w | head -60
