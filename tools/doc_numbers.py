#!/usr/bin/env python
"""The numbers DESIGN.md section 6 / BASELINE.md section 5 / README.md quote, from a bundle under profiles/ (or any directory
with the same file names): one line per workload.  usage: python tools/doc_numbers.py [dir] [prefix]"""
import json
import os
import sys


def lj(p):
    with open(p) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "profiles"
    pre = sys.argv[2] if len(sys.argv) > 2 else "r5_"
    for wl in ("c4", "c2", "c3", "k30", "c4rank", "c5rank"):
        p = os.path.join(d, "%s%s_bench.json" % (pre, wl))
        if not os.path.exists(p):
            continue
        b = lj(p)
        r = b["roofline"]
        h = b["half_iteration_kernel_ms"]
        cpu = b.get("cpu_baseline")
        print("%-7s ms %.2f (no check %.2f)  rows/s %.3e  rows kernel %.2f TB/s (%.3f)  iteration %.3f  cpu %s" % (
            wl, b["ms_per_step"], b.get("ms_per_step_without_check", 0.0), b["value"], r["achieved"] / 1e3, r["frac"], r["iteration_frac"],
            ("%.1e" % cpu["value"]) if cpu else "-"))
        print("        x: rows %.2f ms %.2f TB/s, dual %s ms %s, gramian %.3f | y: rows %.2f ms %.2f TB/s, segments %.2f ms %.2f TB/s, gramian %.3f" % (
            h["x_half"]["rows"], h["x_half"]["rows_GBps"] / 1e3, h["x_half"]["dual"], h["x_half"].get("dual_GBps"), h["x_half"]["gramian"],
            h["y_half"]["rows"], h["y_half"]["rows_GBps"] / 1e3, h["y_half"]["segments"], (h["y_half"]["segments_GBps"] or 0) / 1e3, h["y_half"]["gramian"]))
        if wl == "c4":
            f, u = b["roofline_fp32"], b["roofline_unplanted"]
            print("        fp32: ms %.1f rows/s %.2e rows %.2f (%.3f) it %.3f (+%.0f %%) | unplanted: ms %.1f rows/s %.3e rows %.2f (%.3f) it %.3f" % (
                f["ms_per_step"], f["value"], f["achieved"] / 1e3, f["frac"], f["iteration_frac"], 100 * f["slower_than_split_f16_by"],
                u["ms_per_step"], u["value"], u["achieved"] / 1e3, u["frac"], u["iteration_frac"]))
            print("        traffic %.1f GB vs algorithmic %.1f GB -> frac_counter %.3f; gramian %.3f ms mfma busy %s; launches %s" % (
                (r["traffic"] or 0) / 1e9, r["algorithmic_bytes_per_launch"] / 1e9, r.get("frac_counter") or 0, b["gramian"]["ms_per_step"],
                b["gramian"]["mfma_busy_frac"], r["all_launches_in_process"]))
            print("        cpu sample: %s" % cpu["sample"][:260])
    p = os.path.join(d, pre + "ingest_text_1e9_bench.json")
    if os.path.exists(p):
        i = lj(p)
        print("ingest text: %.1f ms = %.2e lines/s (text kernels %.1f ms = %.0f GB/s of text, copies %.1f, finish %.1f); frac e2e %.3f text %.3f finish %.3f; cpu %.1e" % (
            i["ms"], i["value"], i["text_ms"], i["roofline_text_kernels"]["text_GBps"], i["block_copy_ms"], i["finish_ms"], i["roofline"]["frac"],
            i["roofline_text_kernels"]["frac"], i["roofline_finish"]["frac"], i["cpu_baseline"]["value"]))
    p = os.path.join(d, pre + "ingest_1e9_bench.json")
    if os.path.exists(p):
        i = lj(p)
        print("ingest records: %.1f ms = %.2e records/s (%.3f)" % (i["ms"], i["value"], i["roofline"]["frac"]))
    p = os.path.join(d, pre + "topn_1M_bench.json")
    if os.path.exists(p):
        t = lj(p)
        print("top-N: value %.3e q/s (frac %.3f); batches: %s" % (t["value"], t["roofline"]["frac"], {k: (round(v["queries_per_s"]), round(v["ms_per_call"], 3)) for k, v in t["batches"].items()}))
        for k, v in t.get("by_queries_per_pass", {}).items():
            print("        %s per pass: %.3e q/s, %.1f us per pass, Y frac %.3f" % (k, v["queries_per_s"], v["us_per_pass"], v["Y_stream_frac"]))
        print("        cpu %.2f q/s" % t["cpu_baseline"]["value"])


if __name__ == "__main__":
    main()
