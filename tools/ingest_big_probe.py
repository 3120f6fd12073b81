#!/usr/bin/env python
"""Records (synthetic, generated in HBM piece by piece) -> mals_ingest_finish through the partitioned path
(csrc/ingest_big_host.h): the finish's HIP-event time, the host wall time around it, ranges.  Run under rocprofv3
--kernel-trace --stats for the per-kernel split (profiles/r6_ingest_big_*).
usage: python tools/ingest_big_probe.py [--records N] [--users U] [--items I] [--partition-records P]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=2_500_000_000)
    ap.add_argument("--users", type=int, default=50_000_000)
    ap.add_argument("--items", type=int, default=5_000_000)
    ap.add_argument("--partition-records", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    import torch
    from myrrix_recommender_amd import _lib, ingest
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(1234567890)
    n, chunk = a.records, 250_000_000
    with ingest.Ingest(0) as g:
        g.set_option(_lib.INGEST_OPT_RESERVE_RECORDS, n)
        if a.partition_records:
            g.set_option(_lib.INGEST_OPT_PARTITION_RECORDS, a.partition_records)
        for c0 in range(0, n, chunk):
            m = min(chunk, n - c0)
            u = torch.randint(0, a.users, (m,), device=dev, generator=gen)
            i = (torch.rand(m, device=dev, generator=gen).pow_(3.0) * a.items).long().clamp_(max=a.items - 1)
            v = torch.randint(1, 6, (m,), device=dev, generator=gen).float()
            v[torch.rand(m, device=dev, generator=gen) < 0.01] = float("nan")
            g.append(u, i, v)
            del u, i, v
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        runs = []
        for _ in range(a.repeat):
            t0 = time.perf_counter()
            g.finish()
            wall = (time.perf_counter() - t0) * 1e3
            st = g.stats()
            runs.append({"finish_ms": st["finish_ms"], "host_wall_ms": wall, "workspace_ms": st["workspace_ms"], "radix_passes": st["radix_passes"],
                         "bytes_moved": st["bytes_moved"]})
        c = g.counts()
        free, total = torch.cuda.mem_get_info()
        print(json.dumps({"records": n, "counts": c, "ranges": g.partitions(), "runs": runs, "records_per_s": n / min(r["finish_ms"] for r in runs) * 1e3,
                          "hbm_GB_in_use": round((total - free) / 1e9, 1)}))


if __name__ == "__main__":
    main()
