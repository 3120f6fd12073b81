#!/usr/bin/env python
"""Ingest -> CSR throughput on one MI355X: records/s of mals_ingest_finish on a
synthetic record stream already resident in HBM, the algorithmic bytes all passes move, and the oracle
(record-by-record restatement of the reference's maps, pure Python, 1 core) on a bounded sample.
usage: python tools/bench_ingest.py [--records N] [--users U] [--items I] [--removes P]
       python tools/bench_ingest.py --from-text [...]   the same stream as TEXT ("user,item,value\\n" lines, resident in
           HBM) through mals_ingest_append_text + mals_ingest_finish: lines/s of the whole path, the text kernels'
           own rate, and the end-to-end roofline on the bytes that must move (text in + both CSR matrices out)"""
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
    ap.add_argument("--from-text", action="store_true")
    ap.add_argument("--text-chunk", type=int, default=1 << 25, help="lines formatted per torch pass")
    ap.add_argument("--stream-text", action="store_true",
                    help="--from-text for inputs whose text does not fit beside the ingest (C5: 5e9 lines = 88 GB): every piece of text is "
                         "generated in HBM, handed to mals_ingest_append_text and dropped; times are the library's HIP-event times of its "
                         "kernels (text_info, stats), so the generator between the pieces is not in them; one run, no repeats")
    ap.add_argument("--partition-records", type=int, default=0, help="MALS_INGEST_OPT_PARTITION_RECORDS (0: the library's default)")
    a = ap.parse_args()
    if a.from_text:
        return from_text(a)
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


def format_lines(torch, u, i, v):
    """(user, item, value | NaN) -> the bytes of "user,item,value\\n" lines (value token empty for a remove), on the device:
    fixed-width digit columns + a mask of the columns that print."""
    m = u.shape[0]
    wu, wi = 8, 7
    cols = wu + 1 + wi + 1 + 1 + 1
    mat = torch.empty((m, cols), dtype=torch.uint8, device=u.device)
    keep = torch.ones((m, cols), dtype=torch.bool, device=u.device)
    for j in range(wu):
        p = 10 ** (wu - 1 - j)
        mat[:, j] = ((u // p) % 10 + 48).to(torch.uint8)
        if p > 1:
            keep[:, j] = u >= p
    mat[:, wu] = 44
    for j in range(wi):
        p = 10 ** (wi - 1 - j)
        mat[:, wu + 1 + j] = ((i // p) % 10 + 48).to(torch.uint8)
        if p > 1:
            keep[:, wu + 1 + j] = i >= p
    mat[:, wu + 1 + wi] = 44
    nan = v != v
    mat[:, wu + wi + 2] = (torch.where(nan, torch.zeros_like(v), v).to(torch.int64) + 48).to(torch.uint8)
    keep[:, wu + wi + 2] = ~nan
    mat[:, wu + wi + 3] = 10
    return mat[keep]


def from_text(a):
    import numpy as np
    import torch
    import myrrix_recommender_amd as pkg
    from myrrix_recommender_amd import _lib, ingest
    assert a.users <= 10 ** 8 and a.items <= 10 ** 7
    gen = torch.Generator(device="cuda").manual_seed(1234567890)
    n = a.records
    texts, n_bytes, sample = [], 0, None

    def pieces():
        nonlocal sample
        for lo in range(0, n, a.text_chunk):
            m = min(a.text_chunk, n - lo)
            u = torch.randint(0, a.users, (m,), device="cuda", generator=gen)
            i = (torch.rand(m, device="cuda", generator=gen).pow_(3.0) * a.items).long().clamp_(max=a.items - 1)
            v = torch.randint(1, 6, (m,), device="cuda", generator=gen).float()
            if a.removes > 0:
                v[torch.rand(m, device="cuda", generator=gen) < a.removes] = float("nan")
            t = format_lines(torch, u, i, v)
            if sample is None:
                sample = (u[:200000].cpu().numpy(), i[:200000].cpu().numpy(), v[:200000].cpu().numpy())
            del u, i, v
            yield t, lo + m >= n
    best = None
    if a.stream_text:
        with ingest.Ingest(0) as g:
            g.set_option(_lib.INGEST_OPT_RESERVE_RECORDS, n)
            if a.partition_records:
                g.set_option(_lib.INGEST_OPT_PARTITION_RECORDS, a.partition_records)
            t0 = time.perf_counter()
            for t, last in pieces():
                n_bytes += t.numel()
                g.append_text(t, last)
                del t
            info = g.text_info()
            wall_text = (time.perf_counter() - t0) * 1e3     # (includes the generator: not a rate of anything)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            g.finish()
            st, c = g.stats(), g.counts()
            best = {"total_ms": info["stage_ms"] + info["parse_ms"] + st["finish_ms"], "parse_ms": info["parse_ms"], "stage_ms": info["stage_ms"],
                    "append_text_wall_ms": wall_text, "finish": st, "info": info, "counts": c, "partitions": g.partitions()}
            hbm = torch.cuda.mem_get_info()
            best["hbm_GB_in_use_after_finish"] = round((hbm[1] - hbm[0]) / 1e9, 1)
    else:
        for t, last in pieces():
            texts.append(t)
            n_bytes += t.numel()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    for rep in range(0 if a.stream_text else a.repeat):
        with ingest.Ingest(0) as g:
            g.set_option(_lib.INGEST_OPT_RESERVE_RECORDS, n)
            if a.partition_records:
                g.set_option(_lib.INGEST_OPT_PARTITION_RECORDS, a.partition_records)
            t0 = time.perf_counter()
            for k, t in enumerate(texts):
                g.append_text(t, k == len(texts) - 1)        # one file in len(texts) pieces: lines straddle the pieces
            info = g.text_info()
            t1 = time.perf_counter()
            g.finish()
            st = g.stats()
            c = g.counts()
            wall_text = (t1 - t0) * 1e3
            total_ms = info["stage_ms"] + info["parse_ms"] + st["finish_ms"]
            if best is None or total_ms < best["total_ms"]:
                best = {"total_ms": total_ms, "parse_ms": info["parse_ms"], "stage_ms": info["stage_ms"], "append_text_wall_ms": wall_text, "finish": st, "info": info,
                        "counts": c, "partitions": g.partitions()}
    assert best["info"]["lines"] == n and best["info"]["records"] == n and best["info"]["bad_lines"] == 0
    csr_out = 2.0 * best["counts"]["nnz"] * 8 + 8.0 * (best["counts"]["users"] + best["counts"]["items"] + 2) \
        + 8.0 * (best["counts"]["users"] + best["counts"]["items"])
    # what the text kernels read and write: the text three times (line count, line starts, parse); per line its start
    # offset (4 written, 8 read), status + parsed fields (21 written), status twice more (summary, compaction), the
    # parsed fields once more (20) and the record (20 written)
    parse_bytes = 3.0 * n_bytes + n * (4 + 8 + 21 + 2 + 20 + 20)
    out = {"metric": "ingest lines/s (text of the input files -> records -> two CSR matrices + id tables, on device)",
           "value": n / best["total_ms"] * 1e3, "unit": "lines/s", "ms": best["total_ms"], "text_ms": best["parse_ms"],
           "block_copy_ms": best["stage_ms"],
           "append_text_wall_ms": best["append_text_wall_ms"], "finish_ms": best["finish"]["finish_ms"], "lines": n, "text_bytes": n_bytes,
           "bytes_per_line": n_bytes / n, "users": best["counts"]["users"], "items": best["counts"]["items"], "nnz": best["counts"]["nnz"],
           "full_parser_lines": best["info"]["full_parser_lines"],
           "data": "synthetic, text resident in HBM" + (" piece by piece (generated, appended, dropped: --stream-text)" if a.stream_text else ""),
           "user_ranges": best["partitions"][0], "item_ranges": best["partitions"][1], "hbm_GB_in_use_after_finish": best.get("hbm_GB_in_use_after_finish"),
           "roofline": {"bound": "hbm", "what": "end to end: text bytes in + both CSR matrices and id tables out",
                        "achieved": (n_bytes + csr_out) / best["total_ms"] / 1e6, "peak": 8000.0, "unit": "GB/s",
                        "frac": (n_bytes + csr_out) / best["total_ms"] / 1e6 / 8000.0, "algorithmic_bytes": n_bytes + csr_out},
           "roofline_text_kernels": {"bound": "hbm", "what": "the text kernels alone, on the bytes their passes read and write",
                                     "achieved": parse_bytes / best["parse_ms"] / 1e6, "peak": 8000.0, "unit": "GB/s",
                                     "frac": parse_bytes / best["parse_ms"] / 1e6 / 8000.0, "bytes": parse_bytes,
                                     "text_GBps": n_bytes / best["parse_ms"] / 1e6},
           "roofline_finish": {"bound": "hbm", "achieved": best["finish"]["bytes_moved"] / best["finish"]["finish_ms"] / 1e6, "peak": 8000.0,
                               "unit": "GB/s", "frac": best["finish"]["bytes_moved"] / best["finish"]["finish_ms"] / 1e6 / 8000.0,
                               "radix_passes": best["finish"]["radix_passes"]}}
    if not a.no_cpu_baseline:
        from oracle import ingest_text_oracle as to
        us, is_, vs = sample
        lines = ["%d,%d,%s" % (x, y, "" if z != z else "%d" % z) for x, y, z in zip(us.tolist(), is_.tolist(), vs.tolist())]
        data = ("\n".join(lines) + "\n").encode()
        t0 = time.perf_counter()
        to.expected([data])
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": len(lines) / dt, "unit": "lines/s", "cores": 1, "kind": "port",
                               "sample": "first %d lines, oracle/ingest_text_oracle.py + ingest_oracle.py (pure Python restatement of "
                                         "InputFilesReader.readInputFiles)" % len(lines)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
