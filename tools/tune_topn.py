#!/usr/bin/env python
"""Tuning aid for the top-N pipeline (not part of the product): queries/s of a 4096-query call for the knobs csrc/topn_host.h
reads from the environment (slots in flight, queries per pass, sample size).  usage: python tools/tune_topn.py"""
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import myrrix_recommender_amd as pkg
    rng = np.random.default_rng(1234567890)
    items, n_users, k, how_many = 1_000_000, 100_000, 64, 10
    Y = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
    X = (rng.standard_normal((n_users, k)) / np.sqrt(k)).astype(np.float32)
    deg = 100
    rp = np.arange(n_users + 1, dtype=np.int64) * deg
    col = rng.integers(0, items, n_users * deg).astype(np.int32)
    val = np.ones(n_users * deg, np.float32)
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, n_users)
        core.set_factor_rows(pkg.SIDE_Y, items)
        core.set_factors(pkg.SIDE_X, X)
        core.set_factors(pkg.SIDE_Y, Y)
        core.set_matrix(pkg.SIDE_X, rp, col, val)
        users = rng.integers(0, n_users, 4096).astype(np.int64)
        ref = None
        for slots, per_pass, sample in itertools.product((1, 3, 6), (64, 128, 240), (125000,)):
            os.environ["MALS_TOPN_SLOTS"] = str(slots)
            os.environ["MALS_TOPN_QUERIES_PER_PASS"] = str(per_pass)
            os.environ["MALS_TOPN_SAMPLE_ITEMS"] = str(sample)
            got = core.recommend(users, how_many)
            if ref is None:
                ref = got
            same = all(np.array_equal(a, b) for a, b in zip(ref, got))
            t0 = time.perf_counter()
            for _ in range(10):
                core.recommend(users, how_many)
            dt = (time.perf_counter() - t0) / 10
            passes = (4096 + per_pass - 1) // per_pass
            print(json.dumps({"slots": slots, "per_pass": per_pass, "sample": sample, "ms": round(dt * 1e3, 3), "qps": round(4096 / dt),
                              "us_per_pass": round(dt * 1e6 / passes, 1), "Y_frac": round(passes * items * k * 4 / dt / 8e12, 3), "same_as_first": bool(same)}), flush=True)


if __name__ == "__main__":
    main()
