"""tools/refine_probe.py [workload-ish args]: what the fp32 fast path really loses on bench-like data.
Runs a few iterations, then repeats one half-iteration from the same inputs with refinement off, at the default
limit and with EVERY row refined (limit 1e-30: the fp64-residual answer, used as the reference), and prints the
differences next to the number of rows the default limit marks."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import synth

n_users, n_items, nnz, k, iters = [int(a) for a in sys.argv[1:6]]
alpha = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
lam = float(sys.argv[7]) if len(sys.argv) > 7 else 0.1
dev = torch.device("cuda", 0)
prob = synth.torch_problem(n_users, n_items, nnz, k, dev)
with pkg.ALSCore(k, alpha=alpha, lam=lam) as core:
    core.set_refine_limit(0.0)
    core.set_factor_rows(pkg.SIDE_X, n_users)
    core.set_factor_rows(pkg.SIDE_Y, n_items)
    core.set_matrix(pkg.SIDE_X, *prob["r_csr"])
    core.set_matrix(pkg.SIDE_Y, *prob["c_csr"])
    core.set_factors(pkg.SIDE_Y, prob["Y0"].cpu().numpy())
    for _ in range(iters):
        core.half_iteration(pkg.SIDE_X)
        core.half_iteration(pkg.SIDE_Y)
    core.check()
    for side, name in ((pkg.SIDE_X, "X"), (pkg.SIDE_Y, "Y")):
        res = {}
        for lim in (0.0, None, 1e-30):
            core.set_refine_limit(1024.0 if lim is None else lim)   # None: whatever the default is meant to be
            if lim is None and "MALS_REFINE_LIMIT" in os.environ:
                core.set_refine_limit(float(os.environ["MALS_REFINE_LIMIT"]))
            core.reset_stats()
            core.half_iteration(side)
            core.check()
            res[lim] = (core.get_factors(side).astype(np.float64), core.stats()["rows_refined"])
        ref = res[1e-30][0]
        rms = np.linalg.norm(ref) / np.sqrt(len(ref))
        for lim in (0.0, None):
            F, nref = res[lim]
            d = np.linalg.norm(F - ref, axis=1) / rms
            print("%s half, limit %s: rows refined %d of %d | rel Frobenius vs all-refined %.2e | worst row %.2e | rows above 1e-4: %d" %
                  (name, "default" if lim is None else "off", nref, len(ref), np.linalg.norm(F - ref) / np.linalg.norm(ref), d.max(), int((d > 1e-4).sum())))
        core.set_refine_limit(0.0)
