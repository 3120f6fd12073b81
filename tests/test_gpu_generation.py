"""Generation.recomputeSolver (Generation.java:142-158) -- M^T M on
the GPU (K1) + norm check + MatrixUtils.getSolver -- against the oracle's gramian + RRQR."""
import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import generation
from oracle import oracle

pytestmark = pytest.mark.gpu


def as_map(M, first_id=100):
    return {first_id + 7 * i: M[i] for i in range(M.shape[0])}


@pytest.mark.parametrize("k", [2, 10, 30, 64, 128])
def test_generation_solvers_match_oracle(k):
    rng = np.random.default_rng(10 + k)
    X = rng.standard_normal((700, k)).astype(np.float32)
    Y = rng.standard_normal((300 + k, k)).astype(np.float32)
    gen = generation.Generation(as_map(X), as_map(Y))
    assert gen.getNumUsers() == 700 and gen.getNumItems() == 300 + k
    for solver, M in ((gen.getXTXSolver(), X), (gen.getYTYSolver(), Y)):
        G = oracle.gramian(M)
        for _ in range(3):
            b = rng.standard_normal(k)
            expect = oracle.rrqr_solve(G, b)
            got = solver.solveDToF(b)
            assert np.linalg.norm(got - expect) <= 1e-5 * np.linalg.norm(expect)
            xd = solver.solveFToD(b.astype(np.float32))
            assert np.linalg.norm(xd - expect) <= 1e-5 * np.linalg.norm(expect)


def test_recompute_solver_reports_norm_and_keeps_gramian_installed():
    k = 16
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((1000, k)).astype(np.float32)
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_Y, 1000)
        core.set_factors(pkg.SIDE_Y, Y)
        solver, norm = core.recompute_solver(pkg.SIDE_Y)
        G = oracle.gramian(Y)
        # getNorm() = max abs column sum; 5e-7: the reference rounds each product to fp32 (MU:230), K1 does not
        assert abs(norm - np.abs(G).sum(axis=0).max()) <= 5e-7 * norm
        assert solver.n == k
        # empty side -> no solver (Generation.java:145-147)
        assert core.recompute_solver(pkg.SIDE_X) == (None, 0.0)


def test_ill_conditioned_small_norm():
    """Generation.java:150-153: infNorm < 1 -> IllConditionedSolverException."""
    k = 8
    Y = (1e-3 * np.random.default_rng(2).standard_normal((50, k))).astype(np.float32)
    with pytest.raises(pkg.IllConditionedSolverException):
        generation.Generation({}, as_map(Y))
    assert generation.Generation({}, {}).getYTYSolver() is None


def test_rank_deficient_factors_raise_singular_with_the_oracles_rank():
    k, r = 12, 5
    rng = np.random.default_rng(9)
    Y = (rng.standard_normal((400, r)) @ rng.standard_normal((r, k))).astype(np.float32)
    with pytest.raises(oracle.SingularMatrix) as eo:
        oracle.rrqr_solve(oracle.gramian(Y), np.ones(k))
    with pytest.raises(pkg.SingularMatrixSolverException) as ei:
        generation.Generation({}, as_map(Y))
    assert ei.value.getApparentRank() == eo.value.apparent_rank == r


def test_compute_flags():
    """Generation.java:133-138: model.solver.{xtx,yty}.compute."""
    k = 4
    M = np.random.default_rng(1).standard_normal((40, k)).astype(np.float32)
    pkg.System.setProperty("model.solver.xtx.compute", "false")
    try:
        gen = generation.Generation(as_map(M), as_map(M))
        assert gen.getXTXSolver() is None and gen.getYTYSolver() is not None
    finally:
        pkg.System.clearProperty("model.solver.xtx.compute")
