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
    out = {"metric": "top-N queries/s (all items scored in the reference's arithmetic, known items skipped)", "unit": "queries/s", "items": a.items,
           "features": k, "how_many": a.how_many, "batches": {}}
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, a.users)
        core.set_factor_rows(pkg.SIDE_Y, a.items)
        core.set_factors(pkg.SIDE_X, X)
        core.set_factors(pkg.SIDE_Y, Y)
        core.set_matrix(pkg.SIDE_X, rp, col, val)
        per_pass = 16 * {1: 16, 2: 15, 3: 10, 4: 7}[(k + 31) // 32]        # queries scored per read of Y (csrc/topn_host.h)
        for batch in (1, 16, 64, per_pass, 1024, 4096):
            users = rng.integers(0, a.users, batch).astype(np.int64)
            core.recommend(users, a.how_many)                          # warm
            reps = max(10, 4096 // batch)
            t0 = time.perf_counter()
            for _ in range(reps):
                core.recommend(users, a.how_many)
            dt = (time.perf_counter() - t0) / reps
            passes = (batch + per_pass - 1) // per_pass
            # bytes that must move: Y once per pass (+ the sample of it); everything else (sample rows, candidates) is
            # hundreds of KB
            y_bytes = passes * a.items * k * 4
            out["batches"][str(batch)] = {"ms_per_call": dt * 1e3, "queries_per_s": batch / dt, "passes": passes,
                                          "Y_GBps": y_bytes / dt / 1e9, "Y_stream_frac": y_bytes / dt / 8e12}
        out["queries_per_pass"] = per_pass
        out["value"] = out["batches"]["4096"]["queries_per_s"]
        big = out["batches"]["4096"]
        # the same 4096 queries with fewer queries per read of Y (MALS_TOPN_QUERIES_PER_PASS, the tuning knob of csrc/topn_host.h):
        # more passes, each nearer the speed of one stream of Y
        out["by_queries_per_pass"] = {}
        users = rng.integers(0, a.users, 4096).astype(np.int64)
        for pp in sorted({64, 128, per_pass}):
            if pp > per_pass:
                continue
            os.environ["MALS_TOPN_QUERIES_PER_PASS"] = str(pp)
            core.recommend(users, a.how_many)
            t0 = time.perf_counter()
            for _ in range(10):
                core.recommend(users, a.how_many)
            dt = (time.perf_counter() - t0) / 10
            passes = (4096 + pp - 1) // pp
            out["by_queries_per_pass"][str(pp)] = {"ms_per_call": dt * 1e3, "queries_per_s": 4096 / dt, "passes": passes, "us_per_pass": dt * 1e6 / passes,
                                                   "Y_GBps": passes * a.items * k * 4 / dt / 1e9, "Y_stream_frac": passes * a.items * k * 4 / dt / 8e12}
        del os.environ["MALS_TOPN_QUERIES_PER_PASS"]
        best = max(out["by_queries_per_pass"].values(), key=lambda d: d["Y_stream_frac"])
        out["roofline"] = {"bound": "hbm", "achieved": big["Y_GBps"], "peak": 8000.0, "unit": "GB/s", "frac": big["Y_stream_frac"],
                           "algorithmic_bytes": "items * 4k per pass of %d queries (Y streamed once per pass; passes overlap on six streams)" % per_pass,
                           "at_64_queries_per_call": out["batches"]["64"]["Y_stream_frac"],
                           "at_one_pass_per_call": out["batches"][str(per_pass)]["Y_stream_frac"],
                           "best_over_queries_per_pass": {"frac": best["Y_stream_frac"], "queries_per_s": best["queries_per_s"],
                                                          "queries_per_pass": [int(kk) for kk, v in out["by_queries_per_pass"].items() if v is best][0]}}
        # ---- the way the reference is entered: request threads, one user per call (ServerRecommender.java:359-441) ----------
        # native threads (tools/topn_callers.cpp) on ONE handle; the library folds the concurrent calls into passes
        import ctypes
        harness = os.path.join(ROOT, "tools", "libtopn_callers.so")
        if os.path.exists(harness):
            H = ctypes.CDLL(harness)
            H.topn_callers_run.restype = ctypes.c_int
            H.topn_callers_run.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint64,
                                           ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]
            out["callers"] = {}

            def run_callers(n_threads, calls):
                lat = np.zeros(n_threads * calls, dtype=np.float64)
                wall, chk = ctypes.c_double(0.0), ctypes.c_int64(0)
                before = core.recommend_front_stats()
                rc = H.topn_callers_run(core._h, n_threads, calls, a.users, a.how_many, 42, lat.ctypes.data_as(ctypes.c_void_p), ctypes.byref(wall),
                                        ctypes.byref(chk))
                assert rc == 0, rc
                st = core.recommend_front_stats()
                passes = st["passes"] - before["passes"]
                n = n_threads * calls
                return {"threads": n_threads, "calls": n, "queries_per_s": n / wall.value, "latency_us": {"p50": float(np.percentile(lat, 50)),
                        "p99": float(np.percentile(lat, 99)), "mean": float(lat.mean())}, "passes": passes, "queries_per_pass": n / max(passes, 1),
                        "us_per_pass": wall.value * 1e6 / max(passes, 1), "Y_stream_frac": passes * a.items * k * 4 / wall.value / 8e12}
            for depth in (1, 2, 3):
                core.recommend_set_depth(depth)
                run_callers(8, 50)                                      # warm
                out["callers"]["depth_%d" % depth] = {str(nt): run_callers(nt, 4000 if nt > 1 else 2000) for nt in (1, 8, 32, 128)}
            core.recommend_set_depth(2)
            # waiting callers that poll for their answer before they block (mals_recommend_set_spin_us)
            out["callers_by_spin_us"] = {}
            for spin in (0, 50, 300):
                core.recommend_set_spin_us(spin)
                out["callers_by_spin_us"][str(spin)] = {str(nt): run_callers(nt, 4000) for nt in (8, 32, 128)}
            core.recommend_set_spin_us(0)
            c32 = out["callers"]["depth_2"]["32"]
            out["roofline_callers"] = {"bound": "hbm", "achieved": c32["Y_stream_frac"] * 8000.0, "peak": 8000.0, "unit": "GB/s", "frac": c32["Y_stream_frac"],
                                       "what": "32 native threads of one-user calls on one handle, 2 passes in flight: reads of Y (items * 4k bytes per "
                                               "coalesced pass) per second of wall time", "queries_per_s": c32["queries_per_s"],
                                       "latency_us": c32["latency_us"], "queries_per_pass": c32["queries_per_pass"]}
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
