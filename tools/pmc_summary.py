#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel name.
usage: tools/pmc_summary.py gpurun_out/pmc_<tag> [name-filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "mals::"
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(set))
    each = defaultdict(lambda: defaultdict(dict))   # per dispatch, in dispatch order (the X-half and Y-half launches of one kernel differ)
    for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if filt not in name:
                    continue
                name = name.replace("void ", "")[:60] + " grid=" + row.get("Grid_Size", "?")
                c = row["Counter_Name"]
                agg[name][c] += float(row["Counter_Value"])
                calls[name][c].add(row.get("Dispatch_Id"))
                d = int(row.get("Dispatch_Id") or 0)
                each[name][c][d] = each[name][c].get(d, 0.0) + float(row["Counter_Value"])
    for name in sorted(agg):
        print(name)
        for c in sorted(agg[name]):
            n = max(len(calls[name][c]), 1)
            print("    %-28s per-dispatch %18.1f   (dispatches %d)" % (c, agg[name][c] / n, n))
            if 1 < n <= 16:
                print("        in dispatch order: " + " ".join("%.4g" % each[name][c][d] for d in sorted(each[name][c])))


if __name__ == "__main__":
    main()
