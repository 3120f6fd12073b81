#!/usr/bin/env python
"""Relative Frobenius error of the GPU factors against the fp64 oracle, FP32 vs SPLIT_F16 Gramian."""
import sys
import numpy as np
sys.path.insert(0, ".")
import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import synth, _lib
from oracle import oracle


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))


for k, alpha, vscale in [(64, 1.0, 1.0), (64, 40.0, 1.0), (50, 1.0, 1.0), (64, 1.0, 1000.0), (33, 1.0, 1e-3), (48, 40.0, 50.0)]:
    n_users, n_items, nnz = 20000, 4000, 600000
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, nnz, k, seed=11)
    r_csr = (r_csr[0], r_csr[1], (r_csr[2] * vscale).astype(np.float32))
    c_csr = (c_csr[0], c_csr[1], (c_csr[2] * vscale).astype(np.float32))
    Xo = oracle.half_iteration(*r_csr, Y0, alpha=alpha, threads=8)
    Yo = oracle.half_iteration(*c_csr, Xo, alpha=alpha, threads=8)
    out = []
    for mode in (_lib.GRAMIAN_FP32, _lib.GRAMIAN_SPLIT_F16):
        with pkg.ALSCore(k, alpha=alpha, gramian_mode=mode) as core:
            core.set_factor_rows(pkg.SIDE_X, n_users)
            core.set_factor_rows(pkg.SIDE_Y, n_items)
            core.set_matrix(pkg.SIDE_X, *r_csr)
            core.set_matrix(pkg.SIDE_Y, *c_csr)
            core.set_factors(pkg.SIDE_Y, Y0)
            core.half_iteration(pkg.SIDE_X)
            X = core.get_factors(pkg.SIDE_X)
            core.set_factors(pkg.SIDE_X, Xo)      # same input for the Y half
            core.half_iteration(pkg.SIDE_Y)
            Y = core.get_factors(pkg.SIDE_Y)
            worst = max(rel(X[i], Xo[i].astype(np.float64)) for i in range(0, n_users, 37))
            out.append((rel(X, Xo.astype(np.float64)), rel(Y, Yo.astype(np.float64)), worst))
    print("k=%d alpha=%g vscale=%g  FP32: X %.2e Y %.2e worst-row %.2e | SPLIT_F16: X %.2e Y %.2e worst-row %.2e" %
          ((k, alpha, vscale) + out[0] + out[1]))
