#!/usr/bin/env python
"""Condense a rocprofv3 (rocpd sqlite) kernel trace into the summary committed under profiles/.
usage: tools/rocprof_summary.py results.db [--top N]"""
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 110 else name[:107] + "..."


def main():
    db = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 15
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats summary (top_kernels view; durations in milliseconds)")
    print("%-112s %8s %14s %12s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for n, c, t, a, p in rows[:top]:
        print("%-112s %8d %14.1f %12.1f %6.2f%%" % (short(n), c, t / 1e3, a / 1e3, p))
    print("\n# per-dispatch detail of the mals:: kernels (grid = workgroups x 256 threads)")
    q = ("select name, grid_x, count(*), avg(duration), min(duration), max(duration), max(vgpr_count), "
         "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels "
         "where name like '%mals::%' group by name, grid_x order by avg(duration) desc")
    print("%-70s %12s %6s %12s %12s %12s %5s %5s %5s %6s %7s" %
          ("kernel", "grid_x", "calls", "avg_us", "min_us", "max_us", "vgpr", "agpr", "sgpr", "lds", "scratch"))
    for r in cur.execute(q):
        print("%-70s %12d %6d %12.1f %12.1f %12.1f %5d %5d %5d %6d %7d" %
              (short(r[0])[:70], r[1], r[2], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6], r[7], r[8], r[9], r[10]))


if __name__ == "__main__":
    main()
