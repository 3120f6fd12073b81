#!/usr/bin/env python
"""Error of one half-iteration against the fp64 oracle on inputs with outliers, per solve / Gramian mode
(run on the GPU box).  Prints relative Frobenius error and the worst per-row relative error."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myrrix_recommender_amd as pkg  # noqa: E402
from myrrix_recommender_amd import _lib  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.test_gpu_dual import rows_problem  # noqa: E402


def run(name, k, csr, M, alpha=1.0, lam=0.1):
    Xo = oracle.half_iteration(*csr, M, alpha=alpha, lam=lam, threads=8)
    out = []
    for label, kw in (("direct/fp32", dict(solve_mode=_lib.SOLVE_DIRECT, gramian_mode=_lib.GRAMIAN_FP32)),
                      ("direct/split", dict(solve_mode=_lib.SOLVE_DIRECT, gramian_mode=_lib.GRAMIAN_SPLIT_F16)),
                      ("auto", dict())):
        try:
            with pkg.ALSCore(k, alpha=alpha, lam=lam, **kw) as core:
                core.set_factor_rows(pkg.SIDE_X, len(csr[0]) - 1)
                core.set_factor_rows(pkg.SIDE_Y, M.shape[0])
                core.set_matrix(pkg.SIDE_X, *csr)
                core.set_factors(pkg.SIDE_Y, M)
                core.reset_stats()
                core.half_iteration(pkg.SIDE_X)
                X = core.get_factors(pkg.SIDE_X)
                nd = core.stats()["rows_dual"]
            e = np.linalg.norm(X - Xo) / np.linalg.norm(Xo)
            pr = (np.linalg.norm(X - Xo, axis=1) / np.maximum(np.linalg.norm(Xo, axis=1), 1e-30)).max()
            out.append("%s %.1e/%.1e (dual rows %d)" % (label, e, pr, nd))
        except pkg.MalsError as ex:
            out.append("%s %s" % (label, type(ex).__name__))
    print("%-34s k=%3d  %s" % (name, k, "   ".join(out)), flush=True)


for k in (64, 128):
    nmax = 16 * (k // 32)
    rng = np.random.default_rng(k)
    lengths = np.concatenate([rng.integers(1, nmax + 1, size=300), rng.integers(nmax + 1, 400, size=100)])
    csr, M = rows_problem(lengths, 2000, k, seed=12)
    run("plain", k, csr, M)
    M2 = M.copy(); M2[17] *= 1.0e4
    run("one factor row x1e4", k, csr, M2)
    M2 = M.copy(); M2[17] *= 1.0e2
    run("one factor row x1e2", k, csr, M2)
    v = csr[2].copy(); v[5] *= 1.0e5
    run("one value x1e5", k, (csr[0], csr[1], v), M)
    v = csr[2].copy(); v[::2] *= 1.0e-3; v[1::2] *= 1.0e3 / 5
    run("values 1e-3 and 1e3 mixed", k, (csr[0], csr[1], v), M)
    M2 = (M * np.logspace(0, -3, k)[None, :]).astype(np.float32)
    run("feature scales 1..1e-3", k, csr, M2)
