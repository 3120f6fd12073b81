"""tools/fuzz_probe.py <seed> ...: one fuzz case (tests/test_gpu_fuzz.py) under every arithmetic / solve mode, errors vs the oracle
and vs an exact (numpy fp64) solve of the same systems; per-row worst offenders."""
import sys
import os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import _lib
from oracle import oracle
import test_gpu_fuzz as t


def exact_half(csr, M, alpha, lam, flags):
    rp, col, val = csr
    M = M.astype(np.float64)
    k = M.shape[1]
    G = M.T @ M
    out = np.zeros((len(rp) - 1, k))
    for r in range(len(rp) - 1):
        a, b = rp[r], rp[r + 1]
        y = M[col[a:b]]
        v = val[a:b].astype(np.float64)
        if flags & 1:
            w = np.zeros_like(v); cb = v.copy()
        else:
            w = alpha * np.abs(v); cb = np.where(v > 0, 1 + alpha * np.abs(v), 0.0)
        base = 1.0 if (flags & 2) else 0.0
        W = (0 if (flags & 2) else G) + (y.T * (w + base)) @ y + lam * alpha * (b - a) * np.eye(k)
        rhs = y.T @ cb
        try:
            out[r] = np.linalg.solve(W, rhs)
        except np.linalg.LinAlgError:
            out[r] = np.nan
    return out


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


for seed in map(int, sys.argv[1:]):
    k, n_users, n_items, n_stale, r_csr, c_csr, Y0, cfg = t.draw_case(seed)
    print("seed", seed, "k", k, n_users, n_items, cfg)
    kw = dict(alpha=cfg["alpha"], lam=cfg["lam"], flags=cfg["flags"], threads=4)
    Xo = oracle.half_iteration(*r_csr, Y0, **kw)
    Yo = oracle.half_iteration(*c_csr, Xo, **kw)
    Xe = exact_half(r_csr, Y0, cfg["alpha"], cfg["lam"], cfg["flags"])
    Ye = exact_half(c_csr, Xo, cfg["alpha"], cfg["lam"], cfg["flags"])
    print("  oracle vs exact: X %.2e Y %.2e" % (rel(Xo, Xe), rel(Yo, Ye)))
    for gm in (1, 2):
        for sm in (1, 2):
            c2 = dict(cfg, gramian_mode=gm, solve_mode=sm)
            with pkg.ALSCore(k, **c2) as core:
                core.set_factor_rows(pkg.SIDE_X, n_users)
                core.set_factor_rows(pkg.SIDE_Y, n_items + n_stale)
                core.set_matrix(pkg.SIDE_X, *r_csr)
                core.set_matrix(pkg.SIDE_Y, *c_csr)
                core.set_factors(pkg.SIDE_Y, Y0)
                core.half_iteration(pkg.SIDE_X); core.check()
                X = core.get_factors(pkg.SIDE_X)
                core.set_factors(pkg.SIDE_X, Xo)      # same input as the oracle's second half
                core.half_iteration(pkg.SIDE_Y); core.check()
                Y = core.get_factors(pkg.SIDE_Y)[:n_items]
                st = core.stats()
            d = np.linalg.norm(Y.astype(np.float64) - Ye, axis=1) / max(np.linalg.norm(Ye) / np.sqrt(len(Ye)), 1e-30)
            worst = np.argsort(-d)[:3]
            lens = np.diff(c_csr[0])
            print("  gramian_mode %d solve_mode %d: X vs oracle %.2e exact %.2e | Y vs oracle %.2e exact %.2e | dual rows %d | worst rows %s" %
                  (gm, sm, rel(X, Xo), rel(X, Xe), rel(Y, Yo), rel(Y, Ye), st["rows_dual"], [(int(w), int(lens[w]), float("%.1e" % d[w])) for w in worst]))
