"""Rate of the dual path (als_dual_kernel<T, TN>) per row class: every row of the matrix has the same length L, the
columns are uniform over a 10M-row factor table.  Prints rows/s, entries/s and the algorithmic TB/s ((4k+8) bytes per
entry and per row) for each L -- context for DESIGN.md section 6 (where the k = 128 user half loses its time)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import myrrix_recommender_amd as mra  # noqa: E402

dev = torch.device("cuda", 0)
k = int(os.environ.get("K", 128))
n_cols = int(os.environ.get("COLS", 10_000_000))
n_rows = int(os.environ.get("ROWS", 2_000_000))
lengths = [int(x) for x in os.environ.get("LENGTHS", "8,16,24,32,40,48,56,64").split(",")]
g = torch.Generator(device=dev)
g.manual_seed(7)
Y = torch.empty(n_cols, k, dtype=torch.float32, device=dev).normal_(generator=g).mul_(0.3)
for L in lengths:
    core = mra.ALSCore(k, alpha=1.0, lam=0.1, solve_mode=int(os.environ.get("MODE", 0)))
    core.set_factor_rows(mra.SIDE_Y, n_cols)
    core.bind_factors(mra.SIDE_Y, Y)
    X = torch.zeros(n_rows, k, dtype=torch.float32, device=dev)
    core.set_factor_rows(mra.SIDE_X, n_rows)
    core.bind_factors(mra.SIDE_X, X)
    rp = torch.arange(0, (n_rows + 1) * L, L, dtype=torch.int64, device=dev)
    col = torch.randint(0, n_cols, (n_rows * L,), generator=g, device=dev, dtype=torch.int32)
    val = torch.randint(1, 6, (n_rows * L,), generator=g, device=dev).float()
    core.set_matrix(mra.SIDE_X, rp, col, val)
    core.enable_timing(True)
    for rep in range(3):
        core.reset_stats()
        core.half_iteration(mra.SIDE_X)
        torch.cuda.synchronize()
        st = core.stats()
    ms = st["dual_ms"] if st["rows_dual"] else st["rows_ms"]
    by = (n_rows * L + n_rows) * (4 * k + 8)
    print("L %3d  %s  kernel %.2f ms (rows %.2f dual %.2f rotate %.2f gramian %.2f)  %.2f ns/row  %.2f G entries/s  %.2f TB/s"
          % (L, "dual" if st["rows_dual"] else "rows", ms, st["rows_ms"], st["dual_ms"], st["rotate_ms"], st["gramian_ms"],
             ms * 1e6 / n_rows, n_rows * L / ms / 1e6, by / ms / 1e9), flush=True)
    core.close()
    del X, rp, col, val, core
    torch.cuda.empty_cache()
