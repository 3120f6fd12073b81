#!/usr/bin/env python
"""Ingest -> CSR throughput on one MI355X: records/s of mals_ingest_finish on a
synthetic record stream already resident in HBM, the algorithmic bytes all passes move, and the oracle
(record-by-record restatement of the reference's maps, pure Python, 1 core) on a bounded sample.
usage: python tools/bench_ingest.py [--records N] [--users U] [--items I] [--removes P]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=1_000_000_000)
    ap.add_argument("--users", type=int, default=10_000_000)
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--removes", type=float, default=0.01)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import torch
    import myrrix_recommender_amd as pkg
    from myrrix_recommender_amd import ingest
    gen = torch.Generator(device="cuda").manual_seed(1234567890)
    n = a.records
    u = torch.randint(0, a.users, (n,), device="cuda", generator=gen)
    # items: heavy-tailed popularity like synth.torch_problem
    i = (torch.rand(n, device="cuda", generator=gen).pow_(3.0) * a.items).long().clamp_(max=a.items - 1)
    v = torch.randint(1, 6, (n,), device="cuda", generator=gen).float()
    if a.removes > 0:
        v[torch.rand(n, device="cuda", generator=gen) < a.removes] = float("nan")
    torch.cuda.synchronize()
    torch.cuda.empty_cache()   # hand the generator's temporaries back to the driver before the library allocates
    best, workspace_ms = None, 0.0
    with ingest.Ingest(0) as g:
        g.append(u, i, v)
        for _ in range(a.repeat):
            g.finish()
            st = g.stats()
            workspace_ms = max(workspace_ms, st["workspace_ms"])   # paid once, by the first finish
            if best is None or st["finish_ms"] < best["finish_ms"]:
                best = st
        c = g.counts()
    out = {"metric": "ingest records/s (records -> two CSR matrices + id tables, on device)", "value": n / best["finish_ms"] * 1e3,
           "unit": "records/s", "ms": best["finish_ms"], "records": n, "users": c["users"], "items": c["items"], "nnz": c["nnz"],
           "radix_passes": best["radix_passes"], "workspace_alloc_ms": workspace_ms,
           "roofline": {"bound": "hbm", "achieved": best["bytes_moved"] / best["finish_ms"] / 1e6, "peak": 8000.0, "unit": "GB/s",
                        "frac": best["bytes_moved"] / best["finish_ms"] / 1e6 / 8000.0,
                        "algorithmic_bytes": best["bytes_moved"], "bytes_per_record": best["bytes_moved"] / n}}
    if not a.no_cpu_baseline:
        from oracle import ingest_oracle as io
        m = min(n, 2_000_000)
        us, is_, vs = u[:m].cpu().numpy(), i[:m].cpu().numpy(), v[:m].cpu().numpy()
        t0 = time.perf_counter()
        io.read_input_records(us, is_, vs)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": m / dt, "unit": "records/s", "cores": 1, "kind": "port",
                               "sample": "first %d records, oracle/ingest_oracle.py (pure-Python dict-of-dicts like the reference's map-of-maps)" % m}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
