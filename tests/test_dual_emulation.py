"""The dual solve path as an algorithm (tests/dual_emulation.py, numpy on the CPU) against the oracle, and the
host eigensolver behind it (mals_symmetric_eigen: no GPU needed) against numpy -- the CPU-side half of
tests/test_gpu_dual.py."""
import ctypes

import numpy as np
import pytest

from myrrix_recommender_amd import _lib
from oracle import oracle
from tests import dual_emulation as de


def lib_eigh(G):
    L = _lib.load()
    n = G.shape[0]
    A = np.ascontiguousarray(G, dtype=np.float64)
    ev = np.zeros(n)
    V = np.zeros((n, n))
    rc = L.mals_symmetric_eigen(A.ctypes.data_as(ctypes.c_void_p), n, ev.ctypes.data_as(ctypes.c_void_p), V.ctypes.data_as(ctypes.c_void_p))
    assert rc == _lib.OK
    return ev, V


@pytest.mark.parametrize("n", [1, 2, 3, 10, 30, 50, 64, 100, 128])
def test_host_eigensolver_matches_numpy(n):
    rng = np.random.default_rng(n)
    Y = rng.standard_normal((max(3 * n, 5), n)) * np.logspace(0, -3, n)[None, :]
    if n >= 10:
        Y[:, 3] = Y[:, 2]                      # a repeated direction: rank deficient
    G = Y.T @ Y
    ev, V = lib_eigh(G)
    assert np.allclose(np.sort(ev), np.linalg.eigvalsh(G), rtol=1e-10, atol=1e-12 * np.abs(G).max())
    assert np.linalg.norm(V @ np.diag(ev) @ V.T - G) <= 1e-13 * max(np.linalg.norm(G), 1e-300)
    assert np.linalg.norm(V.T @ V - np.eye(n)) <= 1e-12


def test_host_eigensolver_rejects_bad_input():
    L = _lib.load()
    A = np.array([[1.0, np.nan], [np.nan, 1.0]])
    ev, V = np.zeros(2), np.zeros((2, 2))
    assert L.mals_symmetric_eigen(A.ctypes.data_as(ctypes.c_void_p), 2, ev.ctypes.data_as(ctypes.c_void_p),
                                  V.ctypes.data_as(ctypes.c_void_p)) == _lib.INVALID_ARG
    assert L.mals_symmetric_eigen(None, 2, None, None) == _lib.INVALID_ARG


@pytest.mark.parametrize("k,alpha,lam,scale", [(64, 1.0, 0.1, 1.0), (128, 1.0, 0.1, 1.0), (50, 40.0, 0.001, 1.0), (64, 1.0, 0.1, 1000.0),
                                               (100, 0.01, 10.0, 0.01)])
def test_dual_algorithm_matches_oracle(k, alpha, lam, scale):
    rng = np.random.default_rng(k)
    n_items = 700
    M = rng.standard_normal((n_items, k)).astype(np.float32)
    M /= np.linalg.norm(M, axis=1, keepdims=True).astype(np.float32)
    M = (M * np.logspace(0, -2, k)[None, :]).astype(np.float32)
    lengths = np.concatenate([np.arange(0, 49), rng.integers(1, k // 2 + 1, size=40)])
    row_ptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    col = np.concatenate([np.sort(rng.choice(n_items, size=int(n), replace=False)) for n in lengths]).astype(np.int32)
    val = (rng.integers(1, 6, size=len(col)).astype(np.float32) * np.float32(scale))
    val = np.where(rng.random(len(col)) < 0.15, -val, val).astype(np.float32)
    prep = de.prepare(M, alpha, lam, eigh=lib_eigh)
    w_max_sqrt = np.sqrt(alpha * np.abs(val).max())
    X = np.stack([de.solve_row(prep, col[row_ptr[r]:row_ptr[r + 1]], val[row_ptr[r]:row_ptr[r + 1]], alpha, lam, w_max_sqrt)
                  for r in range(len(lengths))])
    Xo = oracle.half_iteration(row_ptr, col, val, M, alpha=alpha, lam=lam, threads=2)
    err = np.linalg.norm(X - Xo) / np.linalg.norm(Xo)
    per_row = np.linalg.norm(X - Xo, axis=1) / np.maximum(np.linalg.norm(Xo, axis=1), 1e-30)
    assert err < 1e-4 and per_row.max() < 1e-4, (err, per_row.max())   # north-star bar; measured ~3e-7
    assert err < 5e-6
    assert np.all(X[0] == 0)                                           # the empty row
