#!/bin/bash
# kernel trace of one 4096-query top-N call (tuning aid): stats + the timeline of its last dispatches
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/tn
cat > /tmp/tn_one.py <<'P'
import os, sys
sys.path.insert(0, os.environ["ROOT"])
import numpy as np
import myrrix_recommender_amd as pkg
rng = np.random.default_rng(1234567890)
items, n_users, k = 1_000_000, 100_000, 64
Y = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
X = (rng.standard_normal((n_users, k)) / np.sqrt(k)).astype(np.float32)
rp = np.arange(n_users + 1, dtype=np.int64) * 100
col = rng.integers(0, items, n_users * 100).astype(np.int32)
with pkg.ALSCore(k) as core:
    core.set_factor_rows(pkg.SIDE_X, n_users); core.set_factor_rows(pkg.SIDE_Y, items)
    core.set_factors(pkg.SIDE_X, X); core.set_factors(pkg.SIDE_Y, Y)
    core.set_matrix(pkg.SIDE_X, rp, col, np.ones(len(col), np.float32))
    users = rng.integers(0, n_users, 4096).astype(np.int64)
    for _ in range(3):
        core.recommend(users, 10)
P
(cd /tmp; ROOT=$ROOT rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/tn/trace -- python /tmp/tn_one.py > /dev/null 2>&1)
DB=$(find gpurun_out/tn/trace -name "*.db" | head -1)
python tools/rocprof_summary.py $DB --top 20 > gpurun_out/tn/stats.txt 2>&1
python tools/rocprof_timeline.py $DB --last 160 > gpurun_out/tn/timeline.txt 2>&1
rm -rf gpurun_out/tn/trace
