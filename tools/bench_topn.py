#!/usr/bin/env python
"""Top-N scoring on one MI355X: queries/s of mals_recommend for model users
against n_items item vectors resident in HBM, per batch size, + the oracle (numpy restatement of
RecommendIterator + TopN, 1 core) on a few queries.
usage: python tools/bench_topn.py [--items N] [--users U] [--features K] [--how-many N]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--users", type=int, default=100_000)
    ap.add_argument("--features", type=int, default=64)
    ap.add_argument("--how-many", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import myrrix_recommender_amd as pkg
    rng = np.random.default_rng(1234567890)
    k = a.features
    Y = (rng.standard_normal((a.items, k)) / np.sqrt(k)).astype(np.float32)
    X = (rng.standard_normal((a.users, k)) / np.sqrt(k)).astype(np.float32)
    deg = 100
    rp = np.arange(a.users + 1, dtype=np.int64) * deg
    col = rng.integers(0, a.items, a.users * deg).astype(np.int32)
    val = np.ones(a.users * deg, np.float32)
    out = {"metric": "top-N queries/s (all items scored, known items skipped)", "unit": "queries/s", "items": a.items,
           "features": k, "how_many": a.how_many, "batches": {}}
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, a.users)
        core.set_factor_rows(pkg.SIDE_Y, a.items)
        core.set_factors(pkg.SIDE_X, X)
        core.set_factors(pkg.SIDE_Y, Y)
        core.set_matrix(pkg.SIDE_X, rp, col, val)
        for batch in (1, 16, 64, 1024):
            users = rng.integers(0, a.users, batch).astype(np.int64)
            core.recommend(users, a.how_many)                          # warm
            reps = max(1, 2048 // batch)
            t0 = time.perf_counter()
            for _ in range(reps):
                core.recommend(users, a.how_many)
            dt = (time.perf_counter() - t0) / reps
            passes = (batch + 63) // 64
            # bytes the kernels move (filter path): Y once per pass of 64 queries + the 1/16 sample of it; the
            # sample's score rows written once and read by the 4 histogram scans
            moved = passes * a.items * k * 4 * (1 + 1 / 16) + batch * (a.items / 16) * 4 * 5
            out["batches"][str(batch)] = {"ms_per_call": dt * 1e3, "queries_per_s": batch / dt,
                                          "Y_GBps": passes * a.items * k * 4 / dt / 1e9, "moved_GBps": moved / dt / 1e9}
        out["value"] = out["batches"]["1024"]["queries_per_s"]
        out["roofline"] = {"bound": "hbm", "achieved": out["batches"]["1024"]["moved_GBps"], "peak": 8000.0, "unit": "GB/s",
                           "frac": out["batches"]["1024"]["moved_GBps"] / 8000.0,
                           "algorithmic_bytes": "per pass of 64 queries: items*4k*(1+1/16) (Y once + the sample) + 64*(items/16)*4*5 (sample score rows)",
                           "note": "at 64 queries per pass the filter kernel is bound by the fp64 matrix cores (64 MFMAs per 16 items), at 1 query by HBM (kernel: 4.7 TB/s)",
                           "Y_only_frac": out["batches"]["1024"]["Y_GBps"] / 8000.0}
    if not a.no_cpu_baseline:
        from oracle import topn_oracle as to
        t0 = time.perf_counter()
        nq = 5
        for u in range(nq):
            to.recommend(Y, X[u], a.how_many, col[rp[u]:rp[u + 1]])
        out["cpu_baseline"] = {"value": nq / (time.perf_counter() - t0), "unit": "queries/s", "cores": 1, "kind": "port",
                               "sample": "%d queries, oracle/topn_oracle.py (numpy)" % nq}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
