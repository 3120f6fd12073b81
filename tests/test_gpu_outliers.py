"""Robustness envelope of the fp32 / split-f16 arithmetic: outlier values and factor
rows, long (direct kernels) and short (dual kernels) rows together, against the fp64 oracle at the
north-star bar of 1e-4 relative Frobenius.  Per-row errors follow cond(W_u) * 2^-22 -- a row that owns a
1e5x outlier value has an fp32 Cholesky error of ~1e-3 whatever the Gramian arithmetic; the bar is on
the whole factor matrix, as in north_star."""
import os

import numpy as np
import pytest

import myrrix_recommender_amd as pkg  # noqa: F401
from myrrix_recommender_amd import _lib
from oracle import oracle
from tests.test_gpu_dual import dual_max_len, rel, rows_problem, solve_x

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4


def mixed_lengths(k, n_rows, seed):
    nmax = dual_max_len(k)
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.integers(1, nmax + 1, size=n_rows // 2), rng.integers(nmax + 1, 400, size=n_rows - n_rows // 2)])


@pytest.mark.parametrize("mode", [_lib.GRAMIAN_AUTO, _lib.GRAMIAN_FP32])
@pytest.mark.parametrize("k", [64, 128])
def test_one_factor_row_x100(k, mode):
    csr, M = rows_problem(mixed_lengths(k, 600, 1), 2000, k, seed=21)
    M[17] *= 1.0e2
    X, _ = solve_x(k, csr, M, gramian_mode=mode)
    Xo = oracle.half_iteration(*csr, M, threads=4)
    per_row = np.linalg.norm(X - Xo, axis=1) / np.maximum(np.linalg.norm(Xo, axis=1), 1e-30)
    assert rel(X, Xo) < REL_TOL and per_row.max() < REL_TOL, (rel(X, Xo), per_row.max())


@pytest.mark.parametrize("k", [64, 128])
def test_one_value_x1e5(k):
    lengths = mixed_lengths(k, 2000, 2)
    csr, M = rows_problem(lengths, 3000, k, seed=22)
    v = csr[2].copy()
    long_row = int(np.argmax(lengths > dual_max_len(k)))      # a row the direct kernel solves
    v[csr[0][long_row] + 3] = 5.0e5
    v[csr[0][0]] = 5.0e5                                       # and one the dual kernel solves
    csr = (csr[0], csr[1], v)
    X, st = solve_x(k, csr, M)
    Xo = oracle.half_iteration(*csr, M, threads=4)
    assert st["rows_dual"] > 0
    assert rel(X, Xo) < REL_TOL, rel(X, Xo)
    per_row = np.linalg.norm(X - Xo, axis=1) / np.maximum(np.linalg.norm(Xo, axis=1), 1e-30)
    others = np.ones(len(lengths), dtype=bool)
    others[long_row] = False
    assert per_row[others].max() < REL_TOL, per_row[others].max()   # nobody else pays for the outlier
    assert per_row[long_row] < 1e-2                                # cond(W) ~ 1e4 in fp32


@pytest.mark.parametrize("k", [64, 128])
def test_rows_mixing_tiny_and_huge_values(k):
    csr, M = rows_problem(mixed_lengths(k, 600, 3), 2000, k, seed=23)
    v = csr[2].copy()
    v[::2] *= 1.0e-3
    v[1::2] *= 2.0e2
    X, _ = solve_x(k, (csr[0], csr[1], v), M)
    Xo = oracle.half_iteration(csr[0], csr[1], v, M, threads=4)
    per_row = np.linalg.norm(X - Xo, axis=1) / np.maximum(np.linalg.norm(Xo, axis=1), 1e-30)
    assert rel(X, Xo) < REL_TOL and per_row.max() < REL_TOL, (rel(X, Xo), per_row.max())


def test_operand_range_check_switches_to_the_fp32_gather():
    """One value 3e6 times the others over a 200K-row factor matrix stretches bound / typical operand
    (sqrt(w_max / w_mean) * sqrt(max G_ff / mean y_f^2) ~ 2^17) past the 16 binades both f16 halves can
    hold: the launch must run the fp32-gather kernels (bitwise the GRAMIAN_FP32 result), with no host
    round trip (the split kernels return at once on the device-side flag)."""
    k = 64
    lengths = np.random.default_rng(4).integers(40, 300, size=500)
    csr, M = rows_problem(lengths, 200000, k, seed=24)
    v = csr[2].copy()
    v[7] = 1.0e7
    csr = (csr[0], csr[1], v)
    X_auto, _ = solve_x(k, csr, M, solve_mode=_lib.SOLVE_DIRECT)
    X_fp32, _ = solve_x(k, csr, M, solve_mode=_lib.SOLVE_DIRECT, gramian_mode=_lib.GRAMIAN_FP32)
    assert np.array_equal(X_auto, X_fp32)
    # ordinary data: the split kernels run (different bits from the fp32 gather) ...
    csr2, M2 = rows_problem(lengths, 200000, k, seed=24)
    A, _ = solve_x(k, csr2, M2, solve_mode=_lib.SOLVE_DIRECT)
    B, _ = solve_x(k, csr2, M2, solve_mode=_lib.SOLVE_DIRECT, gramian_mode=_lib.GRAMIAN_FP32)
    assert not np.array_equal(A, B) and rel(A, B) < 1e-5
    # ... unless the flag is forced (the switch itself, on the same data)
    os.environ["MALS_FORCE_RANGE_FLAG"] = "0"
    try:
        C, _ = solve_x(k, csr2, M2, solve_mode=_lib.SOLVE_DIRECT)
    finally:
        del os.environ["MALS_FORCE_RANGE_FLAG"]
    assert np.array_equal(C, B)


def test_negative_alpha_runs_the_fp32_gather():
    """alpha < 0 (accepted by the reference, ALS:506-509) has no real sqrt(alpha |r|)."""
    k = 64
    csr, M = rows_problem(np.full(50, 40), 500, k, seed=25, negatives=0.0)
    X, _ = solve_x(k, csr, M, alpha=-0.001)
    Xo = oracle.half_iteration(*csr, M, alpha=-0.001, threads=2)
    assert rel(X, Xo) < REL_TOL
