"""Parity of the HIP path (through the C-ABI) against the oracle and the reference's golden vectors.

Tolerances: the reference computes in fp64 on fp32 storage and pins X*Y^T at 1e-6 absolute on tiny
inputs; the north-star tolerance for the native core is 1e-4 relative Frobenius on the factors.
The HIP path computes the per-row Gramian, Cholesky and solves in fp32 (fp64 only for M^T M), so:
  * factors vs oracle: relative Frobenius <= 1e-4 (asserted; typically ~1e-6),
  * golden X*Y^T:      absolute <= 1e-6, the reference's own tolerance (measured 6e-7 on the fp32 path).
"""
import json
import os

import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import _lib, synth
from oracle import oracle

pytestmark = pytest.mark.gpu

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))
REL_TOL = 1e-4   # north_star: factors within 1e-4 relative Frobenius


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) /
                 max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def make_core(k, n_users, n_y, r_csr, c_csr, Y0, **kw):
    core = pkg.ALSCore(k, **kw)
    core.set_factor_rows(pkg.SIDE_X, n_users)
    core.set_factor_rows(pkg.SIDE_Y, n_y)
    core.set_matrix(pkg.SIDE_X, *r_csr)
    core.set_matrix(pkg.SIDE_Y, *c_csr)
    core.set_factors(pkg.SIDE_Y, Y0)
    return core


# ---- the reference's own unit tests, re-stated through the mirror interface ---------------------
def build_test_xyt_product(case):
    """AlternatingLeastSquaresTest.buildTestXYTProduct (ALST:86-125) / NegativeInputTest (NIT:36-69)."""
    pkg.System.setProperty("model.reconstructRMatrix", "true" if case["flags"] & 1 else "false")
    try:
        byRow, byCol = {}, {}
        R = case["R"]
        for i, row in enumerate(R):
            for j, v in enumerate(row):
                if v != 0:
                    pkg.MatrixUtils.addTo(i, j, float(v), byRow, byCol)
        previousY = {j: np.array(y, dtype=np.float32) for j, y in enumerate(case["Y0"])}
        als = pkg.AlternatingLeastSquares(byRow, byCol, case["features"], case["threshold"],
                                          case["max_iterations"])
        als.setPreviousY(previousY)
        als.call()
        return pkg.MatrixUtils.multiplyXYT(als.getX(), als.getY()), als
    finally:
        pkg.System.clearProperty("model.reconstructRMatrix")


@pytest.mark.parametrize("name", ["als_default", "als_reconstruct_r", "als_negative_input"])
def test_reference_known_answers(name):
    case = GOLDEN[name]
    product, als = build_test_xyt_product(case)
    expected = np.array(case["expected_XYT"], dtype=np.float32)
    assert product.shape == expected.shape
    err = np.max(np.abs(product.astype(np.float32) - expected))
    assert err <= case["tol"], (name, err, als.iterations)   # the reference's own 1e-6 (MyrrixTest.java:34)


def test_call_logs_the_reference_lines_per_iteration(caplog):
    """SURVEY.md section 5 "Metrics/logging": the per-iteration lines of call() (ALS:193, 241-256, 351-358) come out of the
    host mirror while the native loop runs -- through mals_set_iteration_callback, under the reference's logger name --
    with the same convergence values mals_factorize returns."""
    import logging
    case = GOLDEN["als_default"]
    with caplog.at_level(logging.INFO, logger="net.myrrix.online.factorizer.als.AlternatingLeastSquares"):
        _, als = build_test_xyt_product(case)
    text = [r.getMessage() for r in caplog.records]
    assert any(m.startswith("Iterating using 1 GPU(s)") for m in text)
    assert sum(m.startswith("Finished iteration ") for m in text) == als.iterations == len(als.iterationLog)
    assert [i["iteration"] for i in als.iterationLog] == list(range(1, als.iterations + 1))
    assert als.iterationLog[-1]["avg_abs_difference"] == als.convergenceValue
    assert any(m == "Converged" or m == "Reached iteration limit" for m in text)
    assert any("X/tag rows computed" in m and "rows/s" in m for m in text)
    diffs = [m for m in text if m.startswith("Avg absolute difference in estimate vs prior iteration: ")]
    assert len(diffs) >= als.iterations - 1


def test_gramian_known_answer_and_vs_oracle():
    g = GOLDEN["gramian"]
    M = np.array(g["M"], dtype=np.float32)
    with pkg.ALSCore(3) as core:
        core.set_factor_rows(pkg.SIDE_Y, 2)
        core.set_factors(pkg.SIDE_Y, M)
        G = core.gramian(pkg.SIDE_Y, fetch=True)
    assert np.max(np.abs(G - np.array(g["expected_MTM"]))) <= g["tol"]
    rng = np.random.default_rng(3)
    for n, k in [(1, 5), (7, 16), (1000, 30), (4099, 50), (20000, 64), (3001, 100), (513, 128)]:
        M = rng.standard_normal((n, k)).astype(np.float32)
        with pkg.ALSCore(k) as core:
            core.set_factor_rows(pkg.SIDE_X, n)
            core.set_factors(pkg.SIDE_X, M)
            G = core.gramian(pkg.SIDE_X, fetch=True)
        Go = oracle.gramian(M)
        # oracle rounds each product to fp32 first (MU:232); the kernel keeps the exact product
        assert rel(G, Go) < 5e-7, (n, k, rel(G, Go))
        assert np.allclose(G, G.T)


@pytest.mark.parametrize("k", [30, 50, 64, 100, 128])
def test_large_gramian_on_the_f16_pipe_matches_oracle(k):
    """From 262144 rows on, M^T M runs on v_mfma_f32_16x16x16_f16 with split operands, fp32 sums inside 2048-row
    slabs and fp64 sums across them: same bar as the fp64 kernel (5e-7 vs the oracle, whose products are rounded
    to fp32 like MU:232), also with rows of very different magnitudes in one slab."""
    rng = np.random.default_rng(k)
    n = 300_001
    M = rng.standard_normal((n, k)).astype(np.float32)
    M *= np.exp(rng.standard_normal(n) * 2.0).astype(np.float32)[:, None]     # row norms over ~4 decades
    M[12345] *= 1.0e3
    M[200_000:200_016] = 0.0
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, n)
        core.set_factors(pkg.SIDE_X, M)
        G = core.gramian(pkg.SIDE_X, fetch=True)
        G2 = core.gramian(pkg.SIDE_X, fetch=True)
    Go = oracle.gramian(M)
    assert np.array_equal(G, G2)                 # deterministic
    assert np.allclose(G, G.T, rtol=0, atol=0)
    assert rel(G, Go) < 5e-7, (k, rel(G, Go))
    Ge = M.astype(np.float64).T @ M.astype(np.float64)
    assert rel(G, Ge) < 5e-7, (k, rel(G, Ge))   # against exact arithmetic (a handful of rows dominate G here: no averaging over slabs)


# ---- seeded synthetic problems: one half-iteration and full iterations vs the oracle -------------
@pytest.mark.parametrize("k", [1, 2, 10, 16, 30, 33, 50, 64, 100, 128])
def test_half_iterations_match_oracle(k):
    n_users, n_items, nnz = 700, 300, 9000
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, nnz, k, seed=1000 + k, negatives=0.1)
    with make_core(k, n_users, n_items, r_csr, c_csr, Y0) as core:
        core.half_iteration(pkg.SIDE_X)
        X = core.get_factors(pkg.SIDE_X)
        core.half_iteration(pkg.SIDE_Y)
        Y = core.get_factors(pkg.SIDE_Y)
    Xo = oracle.half_iteration(*r_csr, Y0, threads=4)
    Yo = oracle.half_iteration(*c_csr, Xo, threads=4)
    assert np.all(np.isfinite(X)) and np.all(np.isfinite(Y))
    assert rel(X, Xo) < REL_TOL, (k, rel(X, Xo))
    assert rel(Y, Yo) < REL_TOL, (k, rel(Y, Yo))


@pytest.mark.parametrize("k,mode", [(30, 0), (50, 0), (10, 0), (21, _lib.GRAMIAN_FP32), (100, 0)])
def test_padded_gather_table_follows_the_opposite_factors(k, mode):
    """k % 16 != 0: the rows kernels gather from a zero-padded copy of the opposite replica (pad_rows_kernel).
    The copy must follow every change of that replica: chunks solved in any order, the same side solved twice
    in a row after new uploads (with and without a new Gramian: lossIgnoresUnspecified on the fp32 path needs
    none), both sides alternating."""
    n_users, n_items, nnz = 900, 260, 12000
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, nnz, k, seed=4242 + k, negatives=0.1)
    rng = np.random.default_rng(k)
    Y1 = (Y0 + 0.25 * rng.standard_normal(Y0.shape)).astype(np.float32)
    with pkg.ALSCore(k, chunk_rows=250, gramian_mode=mode) as core:
        core.set_factor_rows(pkg.SIDE_X, n_users)
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_matrix(pkg.SIDE_X, *r_csr)
        core.set_matrix(pkg.SIDE_Y, *c_csr)
        n_chunks = core.num_chunks(pkg.SIDE_X)
        assert n_chunks == 4
        got = []
        for Yv, order in ((Y0, range(n_chunks - 1, -1, -1)), (Y1, (2, 0, 3, 1))):
            core.set_factors(pkg.SIDE_Y, Yv)
            core.gramian(pkg.SIDE_Y)
            for c in order:
                core.solve_chunk(pkg.SIDE_X, c)
            core.check()
            got.append(core.get_factors(pkg.SIDE_X))
        core.half_iteration(pkg.SIDE_Y)
        Y2 = core.get_factors(pkg.SIDE_Y)
        core.half_iteration(pkg.SIDE_X)
        X3 = core.get_factors(pkg.SIDE_X)
    X0o = oracle.half_iteration(*r_csr, Y0, threads=4)
    X1o = oracle.half_iteration(*r_csr, Y1, threads=4)
    Y2o = oracle.half_iteration(*c_csr, X1o, threads=4)
    X3o = oracle.half_iteration(*r_csr, Y2o, threads=4)
    for a, b in ((got[0], X0o), (got[1], X1o), (Y2, Y2o), (X3, X3o)):
        assert rel(a, b) < REL_TOL, (k, rel(a, b))
    if mode == _lib.GRAMIAN_FP32:
        # W does not start from G: the same side twice with new uploads and NO new Gramian in between
        flags = pkg.FLAG_LOSS_IGNORES_UNSPECIFIED
        rp = r_csr[0]
        keep = np.nonzero(np.diff(rp) > 0)[0]   # an empty row has W = 0 in this mode: singular in the reference too
        r2 = (np.concatenate([[0], np.cumsum(np.diff(rp)[keep])]).astype(np.int64), r_csr[1], r_csr[2])
        with pkg.ALSCore(k, flags=flags, gramian_mode=mode) as core:
            core.set_factor_rows(pkg.SIDE_X, len(keep))
            core.set_factor_rows(pkg.SIDE_Y, n_items)
            core.set_matrix(pkg.SIDE_X, *r2)
            for Yv in (Y0, Y1):
                core.set_factors(pkg.SIDE_Y, Yv)
                core.solve_side(pkg.SIDE_X)
                core.check()
                X = core.get_factors(pkg.SIDE_X)
                Xo = oracle.half_iteration(*r2, Yv, flags=flags, threads=4)
                assert rel(X, Xo) < REL_TOL, rel(X, Xo)


@pytest.mark.parametrize("flags", [pkg.FLAG_RECONSTRUCT_R, pkg.FLAG_LOSS_IGNORES_UNSPECIFIED,
                                   pkg.FLAG_RECONSTRUCT_R | pkg.FLAG_LOSS_IGNORES_UNSPECIFIED])
def test_mode_flags_match_oracle(flags):
    k, n_users, n_items = 20, 300, 200
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 6000, k, seed=77, negatives=0.2)
    # with lossIgnoresUnspecified an EMPTY row has W = 0 and is singular in the reference too
    # (covered below), so keep only the non-empty rows here
    rp = r_csr[0]
    keep = np.nonzero(np.diff(rp) > 0)[0]
    rp2 = np.concatenate([[0], np.cumsum(np.diff(rp)[keep])]).astype(np.int64)
    r_csr = (rp2, r_csr[1], r_csr[2])
    n_users = len(keep)
    with pkg.ALSCore(k, flags=flags) as core:
        core.set_factor_rows(pkg.SIDE_X, n_users)
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_matrix(pkg.SIDE_X, *r_csr)
        core.set_factors(pkg.SIDE_Y, Y0)
        core.half_iteration(pkg.SIDE_X)
        X = core.get_factors(pkg.SIDE_X)
    Xo = oracle.half_iteration(*r_csr, Y0, flags=flags)
    assert rel(X, Xo) < REL_TOL, (flags, rel(X, Xo))


def test_loss_ignores_unspecified_empty_row_is_singular_like_the_reference():
    k = 4
    Y0 = np.eye(4, dtype=np.float32)
    row_ptr = np.array([0, 2, 2], dtype=np.int64)      # row 1 is empty: W = 0
    col = np.array([0, 1], dtype=np.int32)
    val = np.array([1.0, 2.0], dtype=np.float32)
    with pytest.raises(oracle.SingularMatrix) as eo:
        oracle.half_iteration(row_ptr, col, val, Y0, flags=pkg.FLAG_LOSS_IGNORES_UNSPECIFIED)
    with pkg.ALSCore(k, flags=pkg.FLAG_LOSS_IGNORES_UNSPECIFIED) as core:
        core.set_factor_rows(pkg.SIDE_X, 2)
        core.set_factor_rows(pkg.SIDE_Y, 4)
        core.set_matrix(pkg.SIDE_X, row_ptr, col, val)
        core.set_factors(pkg.SIDE_Y, Y0)
        with pytest.raises(pkg.SingularSystem) as ei:
            core.half_iteration(pkg.SIDE_X)
        assert ei.value.row == 1
        assert ei.value.apparent_rank == eo.value.apparent_rank == 1     # CMLSS:47 getRank(0.01)


@pytest.mark.parametrize("alpha,lam", [(1.0, 0.1), (40.0, 0.1), (1.0, 0.9), (0.5, 0.01)])
def test_alpha_lambda_match_oracle(alpha, lam):
    k, n_users, n_items = 32, 400, 250
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 8000, k, seed=5)
    with make_core(k, n_users, n_items, r_csr, c_csr, Y0, alpha=alpha, lam=lam) as core:
        core.half_iteration(pkg.SIDE_X)
        core.half_iteration(pkg.SIDE_Y)
        X = core.get_factors(pkg.SIDE_X)
        Y = core.get_factors(pkg.SIDE_Y)
    Xo = oracle.half_iteration(*r_csr, Y0, alpha=alpha, lam=lam)
    Yo = oracle.half_iteration(*c_csr, Xo, alpha=alpha, lam=lam)
    assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL, (alpha, lam, rel(X, Xo), rel(Y, Yo))


def test_full_call_matches_oracle_iterations_and_factors():
    k, n_users, n_items = 16, 500, 300
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 10000, k, seed=11)
    tu = np.arange(0, n_users, 5, dtype=np.int64)
    ti = np.arange(0, n_items, 3, dtype=np.int64)
    with make_core(k, n_users, n_items, r_csr, c_csr, Y0) as core:
        iters, conv = core.factorize(0.001, 8, False, tu, ti)
        X = core.get_factors(pkg.SIDE_X)
        Y = core.get_factors(pkg.SIDE_Y)
    Xo, Yo, iters_o, conv_o = oracle.als_call(r_csr, c_csr, n_users, n_items, Y0, k, conv_threshold=0.001,
                                              max_iterations=8, test_users=tu, test_items=ti, threads=4)
    assert iters == iters_o
    assert abs(conv - conv_o) <= 1e-4 * max(abs(conv_o), 1e-6)
    assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL, (rel(X, Xo), rel(Y, Yo))


# ---- edge cases the reference's data model allows ------------------------------------------------
def test_empty_rows_negative_rows_and_stale_y_rows():
    k = 8
    rng = np.random.default_rng(42)
    n_users, n_items, n_stale = 6, 5, 3
    R = np.zeros((n_users, n_items), dtype=np.float32)
    R[0, [0, 2]] = [1, 3]
    R[2, [1]] = [-2]          # negative-only row: contributes to W, nothing to b => x = 0
    R[3, [0, 1, 2, 3, 4]] = [1, 2, 3, 4, 5]
    R[5, [4]] = [2]           # rows 1 and 4 are empty
    r_csr, c_csr = oracle.dense_to_csr(R)
    Y0 = rng.standard_normal((n_items + n_stale, k)).astype(np.float32)   # stale rows count in Y^T Y (N3)
    with make_core(k, n_users, n_items + n_stale, r_csr, c_csr, Y0) as core:
        core.half_iteration(pkg.SIDE_X)
        X = core.get_factors(pkg.SIDE_X)
        core.half_iteration(pkg.SIDE_Y)
        Y = core.get_factors(pkg.SIDE_Y)
    Xo = oracle.half_iteration(*r_csr, Y0)
    assert np.all(X[1] == 0) and np.all(X[4] == 0) and np.all(X[2] == 0)
    assert rel(X, Xo) < REL_TOL
    Yo = Y0.copy()
    Yo[:n_items] = oracle.half_iteration(*c_csr, Xo)
    assert np.array_equal(Y[n_items:], Y0[n_items:])       # stale rows untouched
    assert rel(Y, Yo) < REL_TOL


def test_long_rows_split_into_segments_match_unsplit():
    """Rows longer than segment_nnz take the segments+finish path; same result as the fused path."""
    k, n_users, n_items = 24, 40, 3000
    rng = np.random.default_rng(8)
    lens = [0, 1, 3, 4, 5, 63, 64, 65, 257, 1000, 2999] + list(rng.integers(1, 600, size=n_users - 11))
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    col = np.concatenate([rng.choice(n_items, size=l, replace=False) for l in lens]).astype(np.int32)
    val = rng.integers(1, 6, size=len(col)).astype(np.float32)
    Y0 = (rng.standard_normal((n_items, k)) / np.sqrt(k)).astype(np.float32)
    outs = []
    # (the 2 999-entry row is 47 segments of 64: more than 32 partial slots, summed in groups first -- als_prereduce_kernel;
    # chunk_rows = 7 puts the long rows, their slots and their slot groups into different chunks of the work lists)
    for seg, chunk_rows in ((0, 0), (64, 0), (128, 0), (64, 7)):
        with pkg.ALSCore(k, segment_nnz=seg, chunk_rows=chunk_rows) as core:
            core.set_factor_rows(pkg.SIDE_X, n_users)
            core.set_factor_rows(pkg.SIDE_Y, n_items)
            core.set_matrix(pkg.SIDE_X, row_ptr, col, val)
            core.set_factors(pkg.SIDE_Y, Y0)
            core.half_iteration(pkg.SIDE_X)
            outs.append(core.get_factors(pkg.SIDE_X))
    Xo = oracle.half_iteration(row_ptr, col, val, Y0)
    for X in outs:
        assert rel(X, Xo) < REL_TOL, rel(X, Xo)
    assert rel(outs[1], outs[0]) < 1e-5 and rel(outs[2], outs[0]) < 1e-5
    assert np.array_equal(outs[3], outs[1])     # the same kernels on the same lists, chunk by chunk


def test_chunked_upload_equals_single_upload():
    k, n_users, n_items = 12, 333, 111
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 4000, k, seed=21)
    res = []
    for chunked in (False, True):
        with pkg.ALSCore(k) as core:
            core.set_factor_rows(pkg.SIDE_X, n_users)
            core.set_factor_rows(pkg.SIDE_Y, n_items)
            if chunked:
                core.set_matrix_chunked(pkg.SIDE_X, *r_csr, rows_per_chunk=50)
            else:
                core.set_matrix(pkg.SIDE_X, *r_csr)
            core.set_factors(pkg.SIDE_Y, Y0)
            core.half_iteration(pkg.SIDE_X)
            res.append(core.get_factors(pkg.SIDE_X))
    assert np.array_equal(res[0], res[1])


def test_singular_system_is_reported():
    """lambda = 0 and an empty row with rank-deficient G: the reference throws
    SingularMatrixSolverException from the worker (CMLSS:46-54 via ALS:494)."""
    k = 4
    Y0 = np.zeros((3, k), dtype=np.float32)
    Y0[:, 0] = [1, 2, 3]                      # rank-1 Y => singular Y^T Y
    row_ptr = np.array([0, 1, 1], dtype=np.int64)
    col = np.array([0], dtype=np.int32)
    val = np.array([1.0], dtype=np.float32)
    with pkg.ALSCore(k, lam=0.0) as core:
        core.set_factor_rows(pkg.SIDE_X, 2)
        core.set_factor_rows(pkg.SIDE_Y, 3)
        core.set_matrix(pkg.SIDE_X, row_ptr, col, val)
        core.set_factors(pkg.SIDE_Y, Y0)
        with pytest.raises(pkg.SingularSystem) as ei:
            core.half_iteration(pkg.SIDE_X)
        assert ei.value.side == pkg.SIDE_X and ei.value.row in (0, 1)
        got_row, got_rank = ei.value.row, ei.value.apparent_rank
    # the apparent rank the exception carries (DelegateGenerationManager.java:345-354 lowers
    # model.features to it) is the oracle's for that row
    with pytest.raises(oracle.SingularMatrix) as eo:
        oracle.solve_rows(row_ptr, col, val, Y0, oracle.gramian(Y0), lam=0.0, row_begin=got_row,
                          row_end=got_row + 1)
    assert got_rank == eo.value.apparent_rank == 1


def test_determinism_bitwise():
    k, n_users, n_items = 50, 800, 400
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 20000, k, seed=2)
    outs = []
    for _ in range(2):
        with make_core(k, n_users, n_items, r_csr, c_csr, Y0) as core:
            core.half_iteration(pkg.SIDE_X)
            core.half_iteration(pkg.SIDE_Y)
            outs.append((core.get_factors(pkg.SIDE_X), core.get_factors(pkg.SIDE_Y)))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_mid_size_problem_k64_several_iterations():
    """Scaled-down C4 shape (k=64, power-law items): 3 full iterations stay within tolerance."""
    k, n_users, n_items = 64, 20000, 4000
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 600000, k, seed=64)
    with make_core(k, n_users, n_items, r_csr, c_csr, Y0) as core:
        for _ in range(3):
            core.half_iteration(pkg.SIDE_X)
            core.half_iteration(pkg.SIDE_Y)
        X = core.get_factors(pkg.SIDE_X)
        Y = core.get_factors(pkg.SIDE_Y)
    Xo, Yo = None, Y0
    for _ in range(3):
        Xo = oracle.half_iteration(*r_csr, Yo, threads=8)
        Yo = oracle.half_iteration(*c_csr, Xo, threads=8)
    assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL, (rel(X, Xo), rel(Y, Yo))


# ---- cfg.gramian_mode: fp32 products vs two-f16 split operands (gather_row / gather_row_h) --------
@pytest.mark.parametrize("mode", [_lib.GRAMIAN_FP32, _lib.GRAMIAN_SPLIT_F16])
@pytest.mark.parametrize("k,flags,alpha,vscale", [
    (64, 0, 1.0, 1.0), (50, 0, 40.0, 1.0), (33, 0, 1.0, 1000.0), (48, 0, 1.0, 1e-3), (16, 0, 1.0, 1.0),
    (80, 0, 1.0, 1.0), (96, 0, 40.0, 1.0), (100, 0, 1.0, 1.0), (128, 0, 1.0, 10.0), (2, 0, 1.0, 1.0),
    (10, 0, 40.0, 30.0), (64, pkg.FLAG_RECONSTRUCT_R, 1.0, 1.0),
    (40, pkg.FLAG_LOSS_IGNORES_UNSPECIFIED, 1.0, 1.0),
    (64, pkg.FLAG_RECONSTRUCT_R | pkg.FLAG_LOSS_IGNORES_UNSPECIFIED, 1.0, 1.0)])
def test_gramian_modes_match_oracle(mode, k, flags, alpha, vscale):
    n_users, n_items = 700, 300
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 60000, k, seed=100 + k)
    r_csr = (r_csr[0], r_csr[1], (r_csr[2] * vscale).astype(np.float32))
    keep = np.flatnonzero(np.diff(r_csr[0]) >= (k if flags & 2 else 0))   # lossIgnores: n_u >= k or W is singular
    rp = np.concatenate([[0], np.cumsum(np.diff(r_csr[0])[keep])]).astype(np.int64)
    ent = np.concatenate([np.arange(r_csr[0][i], r_csr[0][i + 1]) for i in keep]) if len(keep) else np.zeros(0, np.int64)
    r_csr = (rp, r_csr[1][ent], r_csr[2][ent])
    assert len(keep) > 50
    with pkg.ALSCore(k, alpha=alpha, flags=flags, gramian_mode=mode) as core:
        core.set_factor_rows(pkg.SIDE_X, len(keep))
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_matrix(pkg.SIDE_X, *r_csr)
        core.set_factors(pkg.SIDE_Y, Y0)
        core.half_iteration(pkg.SIDE_X)
        X = core.get_factors(pkg.SIDE_X)
    Xo = oracle.half_iteration(*r_csr, Y0, alpha=alpha, flags=flags)
    assert rel(X, Xo) < REL_TOL, (mode, k, flags, rel(X, Xo))
    worst = max(rel(X[i], Xo[i]) for i in range(len(keep)))
    assert worst < 10 * REL_TOL, (mode, k, flags, worst)


@pytest.mark.parametrize("k,flags,alpha,vscale", [(64, 0, 1.0, 1.0), (50, 0, 40.0, 1.0), (49, 0, 1.0, 1000.0), (60, 0, 1.0, 1e-3),
                                                  (64, pkg.FLAG_RECONSTRUCT_R, 1.0, 1.0), (56, pkg.FLAG_LOSS_IGNORES_UNSPECIFIED, 1.0, 1.0)])
def test_three_term_split_matches_oracle_no_worse_than_two_terms(k, flags, alpha, vscale):
    """MALS_GRAMIAN_SPLIT3_F16 (features 49..64): every fp32 operand exactly as three f16 terms, six products per tile.  Same
    bar as the other modes; and its error against the fp64 oracle is not above the two-term split's."""
    n_users, n_items = 900, 300
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 70000, k, seed=300 + k)
    r_csr = (r_csr[0], r_csr[1], (r_csr[2] * vscale).astype(np.float32))
    keep = np.flatnonzero(np.diff(r_csr[0]) >= (k if flags & 2 else 0))
    rp = np.concatenate([[0], np.cumsum(np.diff(r_csr[0])[keep])]).astype(np.int64)
    ent = np.concatenate([np.arange(r_csr[0][i], r_csr[0][i + 1]) for i in keep])
    r_csr = (rp, r_csr[1][ent], r_csr[2][ent])
    Xo = oracle.half_iteration(*r_csr, Y0, alpha=alpha, flags=flags)
    err = {}
    for mode in (_lib.GRAMIAN_SPLIT_F16, _lib.GRAMIAN_SPLIT3_F16):
        with pkg.ALSCore(k, alpha=alpha, flags=flags, gramian_mode=mode, solve_mode=_lib.SOLVE_DIRECT) as core:
            core.set_factor_rows(pkg.SIDE_X, len(keep))
            core.set_factor_rows(pkg.SIDE_Y, n_items)
            core.set_matrix(pkg.SIDE_X, *r_csr)
            core.set_factors(pkg.SIDE_Y, Y0)
            core.half_iteration(pkg.SIDE_X)
            X = core.get_factors(pkg.SIDE_X)
        err[mode] = rel(X, Xo)
        assert err[mode] < REL_TOL, (mode, k, flags, err[mode])
    assert err[_lib.GRAMIAN_SPLIT3_F16] <= 1.5 * err[_lib.GRAMIAN_SPLIT_F16] + 1e-8, err
    with pytest.raises(pkg.MalsError):
        pkg.ALSCore(100, gramian_mode=_lib.GRAMIAN_SPLIT3_F16)


def test_split_f16_mode_limits_and_degenerate_scales():
    with pytest.raises(pkg.MalsError):
        pkg.ALSCore(64, gramian_mode=7)
    # all-zero opposite factors (G = 0: scale falls back to 1); huge / tiny factors and values
    # (scale far from 1), combined so that W stays well conditioned in fp32
    k = 64
    r_csr, c_csr, Y0 = synth.numpy_problem(300, 200, 20000, k, seed=5)
    for Y, vs in ((np.zeros_like(Y0), 1.0), (Y0 * 1e4, 1e-4), (Y0 * 1e-6, 1e6), (Y0 * 300, 100.0)):
        rc = (r_csr[0], r_csr[1], (r_csr[2] * vs).astype(np.float32))
        with pkg.ALSCore(k, gramian_mode=_lib.GRAMIAN_SPLIT_F16) as core:
            core.set_factor_rows(pkg.SIDE_X, 300)
            core.set_factor_rows(pkg.SIDE_Y, 200)
            core.set_matrix(pkg.SIDE_X, *rc)
            core.set_factors(pkg.SIDE_Y, Y)
            try:
                core.half_iteration(pkg.SIDE_X)
                X = core.get_factors(pkg.SIDE_X)
                assert np.all(np.isfinite(X))
                Xo = oracle.half_iteration(*rc, Y)
                assert rel(X, Xo) < REL_TOL or np.linalg.norm(Xo) == 0
            except pkg.SingularSystem:
                with pytest.raises(oracle.SingularMatrix):      # singular in the reference too
                    oracle.half_iteration(*rc, Y)
