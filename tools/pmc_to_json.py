#!/usr/bin/env python
"""Turn the PMC passes of tools/profile_round.sh into the two JSON files committed under profiles/:
   <prefix>_pmc_per_dispatch.json  every counter, per kernel, averaged per dispatch
   pmc_traffic.json                HBM-side traffic of the dominant (rows) kernel per launch, read
                                   by bench.py for roofline.traffic
usage: tools/pmc_to_json.py gpurun_out/<tag> profiles/<prefix> <workload> <k>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root, prefix, workload, k = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    tot = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(lambda: defaultdict(set))
    for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if "mals::" not in name:
                    continue
                name = name[:70]
                tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[name][row["Counter_Name"]].add(row.get("Dispatch_Id"))
    out = {}
    for name in tot:
        out[name] = {c: tot[name][c] / max(len(disp[name][c]), 1) for c in tot[name]}
        out[name]["dispatches"] = max(len(v) for v in disp[name].values())
    json.dump(out, open(prefix + "_pmc_per_dispatch.json", "w"), indent=1)
    rows = [n for n in out if "als_persistent_kernel_h" in n and ", 0, " in n] or \
           [n for n in out if "als_persistent_kernel" in n and ", 0, " in n]
    rows = sorted(rows, key=lambda n: -out[n].get("GRBM_GUI_ACTIVE", 0.0))[:1]   # the variant that did the work (its fp32 twin returns at once)
    assert len(rows) == 1, rows
    r = out[rows[0]]
    gram = sorted([n for n in out if "gramian_partial_kernel" in n or "gramian_split_kernel" in n],
                  key=lambda n: -out[n].get("GRBM_GUI_ACTIVE", 0.0) * out[n].get("dispatches", 1))
    read_fetch = 2.0 * 1024.0 * r["FETCH_SIZE"]            # KiB, x2: MI355X_MICROARCH.md (gfx950 reports half)
    read_rdreq = 128.0 * r["TCC_EA0_RDREQ_128B"] + 64.0 * r["TCC_EA0_RDREQ_64B"]
    write = 1024.0 * r["WRITE_SIZE"]
    traffic = {
        "workload": workload, "k": k, "kernel": rows[0], "dispatches_averaged": r["dispatches"],
        "FETCH_SIZE_KB_per_launch": r["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": r["WRITE_SIZE"],
        "TCC_EA0_RDREQ_128B_per_launch": r["TCC_EA0_RDREQ_128B"], "TCC_EA0_RDREQ_64B_per_launch": r["TCC_EA0_RDREQ_64B"],
        "read_bytes_per_launch_fetch_x2": read_fetch, "read_bytes_per_launch_rdreq": read_rdreq,
        "write_bytes_per_launch": write, "traffic_bytes_per_launch": read_fetch + write,
        "method": "rocprofv3 --pmc in separate passes (FETCH_SIZE | WRITE_SIZE + TCC_EA0_RDREQ_{128B,64B}), no tracing "
                  "domains; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes) and cross-checked "
                  "against 128*RDREQ_128B + 64*RDREQ_64B; averaged over the X-half and Y-half launches like roofline.achieved",
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: 1024 SIMDs x (GRBM/8) cycles
        "matrix_pipe_busy_fraction": r["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * r["GRBM_GUI_ACTIVE"] / 8.0),
        "wave_cycles_split": {kk: r[kk] / r["SQ_WAVE_CYCLES"] for kk in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")},
        # K1 (M^T M on the fp64 matrix cores): busy fraction of the matrix pipe over its launches (north_star:
        # "MFMA utilisation on the Gramian")
        "gramian_mfma_busy_fraction": (out[gram[0]]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * out[gram[0]]["GRBM_GUI_ACTIVE"] / 8.0)) if gram else None,
        "gramian_kernel": gram[0] if gram else None,
    }
    json.dump(traffic, open(os.path.join(os.path.dirname(prefix), "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
