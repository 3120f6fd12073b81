"""The committed round bundle and the documents that quote it (CPU): the headline bench line under profiles/ carries every field
of the bench contract, its roofline arithmetic is self-consistent, the kernel-trace summary next to it agrees with it, and
DESIGN.md / BASELINE.md / README.md quote THIS line's numbers (a refreshed bundle with stale tables, or the reverse, fails here)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def last_json(name):
    with open(os.path.join(PROF, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_headline_line_has_the_contract_fields_and_consistent_arithmetic():
    d = last_json("r6_c4_bench.json")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "ms_per_step_without_check", "value_without_check", "roofline_fp32",
                "roofline_unplanted", "roofline_split24", "gather_scale"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["unit"] == "rows/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rows = 10_000_000 + 1_000_000
    assert abs(d["value"] - rows / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    assert r["traffic"] is None or 0.5 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.5     # nothing re-read, nothing missing
    assert 0.0 < r["iteration_frac"] < r["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "rows/s" and "work units" in c["sample"] and c["cpu_model"]
    assert d["ms_per_step"] >= 0.98 * d["ms_per_step_without_check"]            # the headline is the checked region
    assert d["dtype"].startswith("f32 storage; per-row Gramian: f16x2-split")
    f = d["roofline_fp32"]
    assert f["ms_per_step"] > d["ms_per_step"] and abs(f["slower_than_split_f16_by"] - (f["ms_per_step"] / d["ms_per_step"] - 1.0)) < 1e-6
    t3 = d["roofline_split24"]      # three f16 terms per operand: between the two
    assert d["ms_per_step"] < t3["ms_per_step"] < f["ms_per_step"] and f["frac"] < t3["frac"] < d["roofline"]["frac"]


def test_kernel_trace_summary_agrees_with_the_line():
    d = last_json("r6_c4_bench_traced.json")
    tally = d["roofline"]["all_launches_in_process"]
    text = open(os.path.join(PROF, "r6_c4_bench_kernel_stats.txt")).read()
    m = re.search(r"als_persistent_kernel_h<4, 0, true, 2>\(mals::SolveParams\)\s+(\d+)\s+([\d.]+)\s+([\d.]+)", text)
    assert m, "the dominant kernel is in the trace summary"
    calls, avg_ms = int(m.group(1)), float(m.group(3))
    assert calls == tally["launches"]
    assert abs(avg_ms - tally["avg_ms"]) / avg_ms < 0.03, (avg_ms, tally)     # HIP events inside bench.py vs rocprofv3's own clock


def test_documents_quote_this_bundle():
    d = last_json("r6_c4_bench.json")
    ms, frac, itf = "%.1f" % d["ms_per_step"], "%.3f" % d["roofline"]["frac"], "%.3f" % d["roofline"]["iteration_frac"]
    for doc in ("DESIGN.md", "BASELINE.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        assert ms in text and frac in text and itf in text, (doc, ms, frac, itf)
    for wl in ("c5rank", "c2", "c3", "k30", "c4rank"):
        w = last_json("r6_%s_bench.json" % wl)
        q = ("%.1f" if w["ms_per_step"] >= 100 else "%.2f" if w["ms_per_step"] < 20 else "%.1f") % w["ms_per_step"]
        for doc in ("DESIGN.md", "BASELINE.md"):
            assert q in open(os.path.join(ROOT, doc)).read(), (doc, wl, q)


def test_the_next_rows_carry_roofline_and_cpu_baseline_too():
    t = last_json("r6_topn_1M_bench.json")
    assert t["unit"] == "queries/s" and t["value"] == t["batches"]["4096"]["queries_per_s"]
    r = t["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert set(t["by_queries_per_pass"]) >= {"64", "128"} and r["best_over_queries_per_pass"]["frac"] >= r["frac"] * 0.95
    assert t["cpu_baseline"]["kind"] == "port" and t["cpu_baseline"]["value"] > 0
    stats = open(os.path.join(PROF, "r6_topn_1M_kernel_stats.txt")).read()
    assert "topn_stream_kernel" in stats and "topn_rescore_kernel" in stats
    for name, unit in (("r6_ingest_text_1e9_bench.json", "lines/s"), ("r6_ingest_1e9_bench.json", "records/s")):
        i = last_json(name)
        assert i["unit"] == unit and i["value"] > 1e9 and 0.0 < i["roofline"]["frac"] < 1.0
        assert i["cpu_baseline"]["kind"] == "port" and i["cpu_baseline"]["value"] > 0
    i = last_json("r6_ingest_text_1e9_bench.json")
    assert abs(i["value"] - i["lines"] / (i["ms"] * 1e-3)) / i["value"] < 1e-6 and i["full_parser_lines"] >= 0


def test_the_round_6_rows_are_in_the_bundle():
    t = last_json("r6_topn_1M_bench.json")
    c = t["callers"]["depth_2"]["32"]
    assert c["threads"] == 32 and c["queries_per_s"] > 5e4 and c["latency_us"]["p50"] < c["latency_us"]["p99"] and c["queries_per_pass"] > 1.5
    rc = t["roofline_callers"]
    assert rc["bound"] == "hbm" and abs(rc["frac"] - c["Y_stream_frac"]) < 1e-9 and rc["frac"] >= 0.30      # VERDICT r5's target for one-user callers
    i = last_json("r6_ingest_text_5e9_bench.json")
    assert i["lines"] == 5_000_000_000 and i["unit"] == "lines/s" and i["value"] > 1e9 and i["user_ranges"] >= 2 and i["item_ranges"] >= 2
    assert abs(i["value"] - i["lines"] / (i["ms"] * 1e-3)) / i["value"] < 1e-6 and i["nnz"] > 2 ** 32 and i["cpu_baseline"]["kind"] == "port"
    assert 0.0 < i["roofline"]["frac"] < 1.0 and i["hbm_GB_in_use_after_finish"] < 288
    stats = open(os.path.join(PROF, "r6_ingest_big_2p5e9_kernel_stats.txt")).read()
    assert "big_select_items_kernel" in stats and "big_compact_part_kernel" in stats and "rs_scatter_kernel" in stats


def test_documents_quote_the_next_rows_of_this_bundle():
    i = last_json("r6_ingest_text_1e9_bench.json")
    ms = "%.0f ms" % i["ms"]
    for doc in ("DESIGN.md", "BASELINE.md", "README.md"):
        assert ms in open(os.path.join(ROOT, doc)).read(), (doc, ms)
    t = last_json("r6_topn_1M_bench.json")
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    for pp in ("64", "128"):
        v = t["by_queries_per_pass"][pp]
        assert ("%.2e" % v["queries_per_s"]).replace("e+0", "e") in text, (pp, v["queries_per_s"])
    assert ("%.2e" % t["value"]).replace("e+0", "e") in text
