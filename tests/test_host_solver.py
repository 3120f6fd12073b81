"""SURVEY.md section 8(f) row 1, host half: MatrixUtils.getSolver / Solver.solveDToF / solveFToD
(MU:137, CMLSS:37-55, CommonsMathSolver.java:37-59) behind mals_solver_*.  Pure host fp64 code in
the product library, checked against the oracle's restatement of commons-math3 3.2 RRQR."""
import time

import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import generation
from oracle import oracle


def random_symmetric(n, rng):
    """SolverLoadTest.java:56-70: uniform [0,1) entries, symmetric."""
    A = rng.random((n, n))
    return np.triu(A) + np.triu(A, 1).T


@pytest.mark.parametrize("k", [1, 2, 3, 10, 30, 64, 100, 128])
def test_solver_matches_oracle_on_gramians(k):
    rng = np.random.default_rng(100 + k)
    M = rng.standard_normal((4 * k + 10, k)).astype(np.float32)
    G = oracle.gramian(M)
    s = generation.getSolver(G)
    for _ in range(3):
        b = rng.standard_normal(k)
        expect = oracle.rrqr_solve(G, b)                    # float32, like solveDToF
        got = s.solveDToF(b)
        assert got.dtype == np.float32
        assert np.allclose(got, expect, rtol=1e-6, atol=1e-7)
        bf = b.astype(np.float32)
        xd = s.solveFToD(bf)
        assert xd.dtype == np.float64
        assert np.allclose(G @ xd, bf.astype(np.float64), rtol=0, atol=1e-9 * max(1.0, np.abs(G).max()))


def test_solver_on_non_spd_symmetric_matrix():
    """getSolver is a general (pivoted QR) solver, not a Cholesky: SolverLoadTest's matrix is
    symmetric indefinite."""
    rng = np.random.default_rng(7)
    A = random_symmetric(60, rng)
    b = rng.random(60)
    x = generation.getSolver(A).solveFToD(b.astype(np.float32))
    assert np.allclose(A @ x, b.astype(np.float32), atol=1e-9)
    assert np.allclose(generation.getSolver(A).solveDToF(b), oracle.rrqr_solve(A, b), rtol=1e-5, atol=1e-6)


def test_singular_matrix_reports_apparent_rank_like_the_oracle():
    rng = np.random.default_rng(3)
    for k, r in [(5, 1), (8, 3), (30, 12), (64, 63)]:
        B = rng.standard_normal((k, r))
        A = B @ B.T                                          # rank r
        with pytest.raises(pkg.SingularMatrixSolverException) as ei:
            generation.getSolver(A)
        with pytest.raises(oracle.SingularMatrix) as eo:
            oracle.rrqr_solve(A, np.ones(k))
        assert ei.value.getApparentRank() == eo.value.apparent_rank == r
        assert not generation.isNonSingular(A)
    assert generation.isNonSingular(np.eye(4))
    assert generation.getSolver(None) is None               # CMLSS:38-40


def test_threshold_property_is_honoured():
    A = np.diag([1.0, 1e-3])
    assert generation.isNonSingular(A)
    pkg.System.setProperty("common.matrix.singularityThreshold", "0.01")
    try:
        assert not generation.isNonSingular(A)
    finally:
        pkg.System.clearProperty("common.matrix.singularityThreshold")


def test_solver_load():
    """SolverLoadTest.java:41-53: getSolver of a 500 x 500 symmetric matrix in < 300 ms each."""
    A = random_symmetric(500, np.random.default_rng(1234567890))
    generation.getSolver(np.eye(2))                          # library load is not part of the timing
    iterations = 5
    t0 = time.perf_counter()
    for _ in range(iterations):
        s = generation.getSolver(A)
    elapsed_ms = (time.perf_counter() - t0) * 1e3
    assert elapsed_ms < 300 * iterations
    b = np.ones(500, dtype=np.float32)
    assert np.allclose(A @ s.solveFToD(b), b, atol=1e-8)
