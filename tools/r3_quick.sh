#!/bin/bash
# usage: tools/r3_quick.sh <tag> "<pytest -k expr or empty>" [bench args...]
set -u
TAG=$1; KEXPR=$2; shift 2
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu -k "$KEXPR" > $OUT/test.log 2>&1; echo "test rc=$?"; tail -15 $OUT/test.log
fi
if [ $# -gt 0 ]; then
  timeout 900 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 2500 $OUT/bench.json; tail -3 $OUT/bench.err
fi
