#!/usr/bin/env python
"""Top-N throughput at other shapes than the bench's (feature counts with each load mode of topn_stream_kernel, a 10M-item
catalogue): queries/s of 4096-query calls.  usage: python tools/topn_shapes.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import myrrix_recommender_amd as pkg
rng = np.random.default_rng(5)
for k, items in ((128, 1_000_000), (100, 1_000_000), (30, 1_000_000), (16, 1_000_000), (64, 10_000_000)):
    Y = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
    X = (rng.standard_normal((20000, k)) / np.sqrt(k)).astype(np.float32)
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, len(X)); core.set_factor_rows(pkg.SIDE_Y, items)
        core.set_factors(pkg.SIDE_X, X); core.set_factors(pkg.SIDE_Y, Y)
        users = rng.integers(0, len(X), 4096).astype(np.int64)
        core.recommend(users, 10, consider_known_items=True)
        t0 = time.perf_counter()
        for _ in range(5):
            core.recommend(users, 10, consider_known_items=True)
        dt = (time.perf_counter() - t0) / 5
        per_pass = 16 * {1: 16, 2: 15, 3: 10, 4: 7}[(k + 31) // 32]
        passes = (4096 + per_pass - 1) // per_pass
        print(json.dumps({"k": k, "items": items, "qps": round(4096 / dt), "us_per_pass": round(dt * 1e6 / passes, 1), "Y_frac": round(passes * items * k * 4 / dt / 8e12, 3)}), flush=True)
