#!/usr/bin/env python
"""Timeline of the last N kernel dispatches of a rocprofv3 (rocpd sqlite) kernel trace: start offset, duration, queue/stream,
kernel -- to see what overlaps what.  usage: tools/rocprof_timeline.py results.db [--last N] [--like PATTERN]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 120
    like = sys.argv[sys.argv.index("--like") + 1] if "--like" in sys.argv else "%"
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(cur.execute("select start, end, %s, name, grid_x from kernels where name like ? order by start desc limit ?" % qcol, (like, last)))
    rows.reverse()
    t0 = rows[0][0]
    print("# columns: %s" % cols)
    print("%10s %9s %6s %10s  %s" % ("start_us", "dur_us", "queue", "grid_x", "kernel"))
    for s, e, q, n, g in rows:
        n = n.replace("void ", "").replace("mals::", "")
        print("%10.1f %9.1f %6s %10d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, g, n[:60]))


if __name__ == "__main__":
    main()
