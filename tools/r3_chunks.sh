#!/bin/bash
# price the chunking of the multi-GPU path on one GPU: the group code path with a one-rank RCCL communicator
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/chunks
for ch in 1 4 8 16; do
  MALS_FORCE_COLLECTIVES=1 timeout 600 python bench.py --no-cpu-baseline --no-unplanted --steps 5 --warmup 2 --exchange-chunks $ch > gpurun_out/chunks/c4_$ch.json 2> gpurun_out/chunks/c4_$ch.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/chunks/c4_$ch.json").read().strip().splitlines()[-1])
print("chunks $ch: ms/step %.2f kernels %s" % (d["ms_per_step"], {k: round(v,2) for k,v in d["kernels_ms_per_step"].items()}))
PY
done
