#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
tools/profile_round.sh r3c4 > gpurun_out/r3c4.log 2>&1
DB=$(find gpurun_out/r3c4/trace -name "*.db" | head -1)
python tools/rocprof_summary.py $DB > gpurun_out/r3c4/bench_kernel_stats.txt 2>&1
rm -rf gpurun_out/r3c4/trace
python tools/pmc_to_json.py gpurun_out/r3c4 gpurun_out/r3c4/r3_c4 c4 64 > gpurun_out/r3c4/pmc_to_json.log 2>&1
find gpurun_out/r3c4 -name "*.csv" -delete
python - <<'PY'
import json
for f in ("bench.json","bench_traced.json"):
    d=json.loads(open("gpurun_out/r3c4/"+f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["all_launches_in_process"])
PY
grep "persistent_kernel_h<4, 0" gpurun_out/r3c4/bench_kernel_stats.txt | head -2
