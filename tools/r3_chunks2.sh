#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/chunks
for ch in 1 2 4 8; do
  MALS_FORCE_COLLECTIVES=1 timeout 600 python bench.py --workload c4shard8 --no-cpu-baseline --no-unplanted --steps 10 --warmup 3 --exchange-chunks $ch > gpurun_out/chunks/s8_$ch.json 2> gpurun_out/chunks/s8_$ch.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/chunks/s8_$ch.json").read().strip().splitlines()[-1])
print("c4shard8 chunks $ch: ms/step %.3f kernels %s" % (d["ms_per_step"], {k: round(v,2) for k,v in d["kernels_ms_per_step"].items()}))
PY
done
