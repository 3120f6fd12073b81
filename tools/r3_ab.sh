#!/bin/bash
# usage: tools/r3_ab.sh <tag> "<workloads>" "<lib names under csrc, or 'default'>" [bench args]
set -u
TAG=$1; WLS=$2; LIBS=$3; shift 3
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for wl in $WLS; do
  for lib in $LIBS; do
    # lib spec: <name>[@VAR=value[,VAR=value]]  (name "default" = the product library)
    spec=$lib; envs=""; case "$lib" in *@*) spec=${lib%%@*}; envs=${lib#*@};; esac
    if [ "$spec" = "default" ]; then unset MALS_LIB; else export MALS_LIB=$ROOT/myrrix-recommender_amd/csrc/$spec; fi
    for kv in $(echo "$envs" | tr ',' ' '); do export "$kv"; done
    timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-unplanted --no-fp32-leg --steps 5 --warmup 2 "$@" > $OUT/${wl}_${lib}.json 2> $OUT/${wl}_${lib}.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/${wl}_${lib}.json").read().strip().splitlines()[-1])
    print("$wl $lib ms/step %.3f rows_frac %.3f iter_frac %.3f kernels %s halves %s" % (d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["iteration_frac"], {k: round(v,2) for k,v in d["kernels_ms_per_step"].items()}, {h: {k: v for k, v in hv.items() if k in ("rows","segments","dual")} for h, hv in d["half_iteration_kernel_ms"].items()}))
except Exception as e:
    print("$wl $lib FAILED", e); print(open("$OUT/${wl}_${lib}.err").read()[-1500:])
PY
    for kv in $(echo "$envs" | tr ',' ' '); do unset "${kv%%=*}"; done
  done
done
