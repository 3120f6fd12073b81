#!/bin/bash
# round 4: the new tests + the chunked exchange A/B (one vs two compute streams) + c4rank / c5rank, one box
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4g
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_jni_sequence.py tests/test_gpu_group.py tests/test_gpu_group_transport.py tests/test_gpu_parity.py -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
for spec in "1:" "4:" "4:--one-compute-stream" "1:" "4:" "4:--one-compute-stream" "8:" "8:--one-compute-stream"; do
  ch=${spec%%:*}; extra=${spec#*:}
  MALS_FORCE_COLLECTIVES=1 timeout 600 python bench.py --no-cpu-baseline --no-unplanted --no-fp32-leg --steps 5 --warmup 2 --exchange-chunks $ch $extra > $OUT/c4_${ch}_${extra}.json 2> $OUT/c4_${ch}_${extra}.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/c4_${ch}_${extra}.json").read().strip().splitlines()[-1])
    print("chunks $ch $extra: ms/step %.2f with check %.2f kernels %s" % (d["ms_per_step"], d["ms_per_step_with_check"], {k: round(v,2) for k,v in d["kernels_ms_per_step"].items()}))
except Exception as e:
    print("chunks $ch $extra FAILED", e); print(open("$OUT/c4_${ch}_${extra}.err").read()[-800:])
PY
done
bash tools/r3_ab.sh r4g "c4rank" "default@MALS_DUAL_PAIRS=0 default@MALS_DUAL_PAIRS=1 default@MALS_DUAL_PAIRS=0 default@MALS_DUAL_PAIRS=1"
