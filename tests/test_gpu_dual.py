"""The dual ("short row") solve path of csrc/dual_kernels.h against the oracle, through the C-ABI.

A row with n_u entries is solved from the n_u x n_u system of the push-through identity in the
eigenbasis of the shared Gramian instead of the k x k system of ALS:447-494 -- the same x_u, so the
bar is the same as for the direct path: 1e-4 relative Frobenius against the fp64 oracle (measured
2-5e-7).  The tests check that the dual kernels really ran (stats.rows_dual) and that every
disqualifying condition falls back to the direct kernels with unchanged results."""
import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import _lib
from oracle import oracle

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) /
                 max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def dual_max_len(k):
    """Largest row the dual path takes at k features: 16 * dual_max_blocks(T) (csrc/dual_kernels.h)."""
    T = (k + 15) // 16
    return 16 * (4 if T >= 6 else (3 if T >= 4 else T // 2))


def rows_problem(lengths, n_items, k, seed, negatives=0.1, vscale=1.0, unit=True):
    """CSR with the given row lengths (distinct random columns per row), values +-1..5, and M (n_items x k)."""
    rng = np.random.default_rng(seed)
    lengths = np.asarray(lengths, dtype=np.int64)
    row_ptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    col = np.concatenate([np.sort(rng.choice(n_items, size=int(n), replace=False)) for n in lengths] or [np.zeros(0)]).astype(np.int32)
    val = rng.integers(1, 6, size=len(col)).astype(np.float32) * np.float32(vscale)
    val = np.where(rng.random(len(col)) < negatives, -val, val).astype(np.float32)
    M = rng.standard_normal((n_items, k)).astype(np.float32)
    if unit:
        M /= np.linalg.norm(M, axis=1, keepdims=True).astype(np.float32)
    return (row_ptr, col, val), M


def solve_x(k, csr, M, refine_limit=None, **kw):
    n_rows = len(csr[0]) - 1
    with pkg.ALSCore(k, **kw) as core:
        if refine_limit is not None:
            core.set_refine_limit(refine_limit)
        core.set_factor_rows(pkg.SIDE_X, n_rows)
        core.set_factor_rows(pkg.SIDE_Y, M.shape[0])
        core.set_matrix(pkg.SIDE_X, *csr)
        core.set_factors(pkg.SIDE_Y, M)
        core.reset_stats()
        core.half_iteration(pkg.SIDE_X)
        return core.get_factors(pkg.SIDE_X), core.stats()


@pytest.mark.parametrize("k", [20, 32, 33, 48, 50, 64, 80, 96, 100, 112, 127, 128])
def test_every_row_length_matches_oracle(k):
    nmax = dual_max_len(k)
    # every length 0 .. nmax + 8 (the lengths above nmax and the empty rows take the direct kernel), three of each
    lengths = np.repeat(np.arange(0, nmax + 9), 3)
    np.random.default_rng(k).shuffle(lengths)
    csr, M = rows_problem(lengths, 500, k, seed=k)
    X, st = solve_x(k, csr, M, solve_mode=_lib.SOLVE_DUAL)
    Xo = oracle.half_iteration(*csr, M, threads=4)
    assert st["rows_dual"] == 3 * nmax, st
    assert st["rows_solved"] == len(lengths)
    assert np.all(np.isfinite(X))
    assert rel(X, Xo) < REL_TOL, (k, rel(X, Xo))
    per_row = np.linalg.norm(X - Xo, axis=1) / np.maximum(np.linalg.norm(Xo, axis=1), 1e-30)
    assert per_row.max() < REL_TOL, (k, int(per_row.argmax()), per_row.max())
    assert np.all(X[lengths == 0] == 0.0)
    # and the same rows through the direct kernels
    Xd, std = solve_x(k, csr, M, solve_mode=_lib.SOLVE_DIRECT)
    assert std["rows_dual"] == 0
    assert rel(X, Xd) < 1e-5


@pytest.mark.parametrize("alpha,lam,vscale", [(40.0, 0.1, 1.0), (1.0, 0.001, 1.0), (40.0, 0.0001, 1.0),
                                              (1.0, 0.1, 1000.0), (1.0, 0.1, 0.001), (0.01, 10.0, 1.0)])
@pytest.mark.parametrize("k", [64, 128])
def test_alpha_lambda_and_value_scales(k, alpha, lam, vscale):
    lengths = np.random.default_rng(5).integers(1, dual_max_len(k) + 1, size=400)
    csr, M = rows_problem(lengths, 3000, k, seed=int(alpha * 10 + k), negatives=0.2, vscale=vscale)
    X, st = solve_x(k, csr, M, alpha=alpha, lam=lam)
    Xo = oracle.half_iteration(*csr, M, alpha=alpha, lam=lam, threads=4)
    assert st["rows_dual"] == len(lengths)
    assert rel(X, Xo) < REL_TOL, (k, alpha, lam, vscale, rel(X, Xo))


@pytest.mark.parametrize("k", [64, 128])
def test_ill_conditioned_gramian(k):
    """Feature scales over six decades (cond(G) ~ 1e12): the rotation is then essential, not cosmetic."""
    lengths = np.random.default_rng(6).integers(1, dual_max_len(k) + 1, size=300)
    csr, M = rows_problem(lengths, 4000, k, seed=11, unit=False)
    M = (M * np.logspace(0, -3, k)[None, :]).astype(np.float32)
    X, st = solve_x(k, csr, M)
    Xo = oracle.half_iteration(*csr, M, threads=4)
    assert st["rows_dual"] == len(lengths)
    assert rel(X, Xo) < REL_TOL, rel(X, Xo)


def test_factor_outlier_rows():
    """One factor row 1e4 times the others.  The reference rounds every product of M^T M to fp32 before
    adding it (MU:232); with this row in, those roundings are ~0.06 absolute on entries whose remaining
    structure is ~0.5, so the reference's OWN result moves by ~1e-2 against exact arithmetic -- there is no
    1e-4 parity to have with it.  What can be checked: the dual path (fp64 rotation, S well conditioned)
    reproduces the exact-arithmetic solution of the same systems to 1e-4."""
    k = 64
    lengths = np.random.default_rng(7).integers(1, 33, size=300)
    csr, M = rows_problem(lengths, 2000, k, seed=12)
    M[17] *= 1.0e4
    X, st = solve_x(k, csr, M)
    assert st["rows_dual"] == len(lengths)
    G_exact = M.astype(np.float64).T @ M.astype(np.float64)
    X_exact = oracle.solve_rows(*csr, M, G_exact, threads=4)
    per_row = np.linalg.norm(X - X_exact, axis=1) / np.maximum(np.linalg.norm(X_exact, axis=1), 1e-30)
    assert rel(X, X_exact) < REL_TOL and per_row.max() < REL_TOL, (rel(X, X_exact), per_row.max())
    X_ref = oracle.half_iteration(*csr, M, threads=4)
    assert rel(X_ref, X_exact) > 1e-3   # the reference itself is that far from exact arithmetic here


def test_rank_deficient_gramian_falls_back_to_direct():
    """lambda = 0 and fewer factor rows than features: G + lambda alpha n I is singular, the dual path
    must not be taken; rows with >= 1 entry still have W = G + update and go through the direct kernel."""
    k = 64
    lengths = np.full(50, 20)
    csr, M = rows_problem(lengths, 40, k, seed=13)
    with pkg.ALSCore(k, lam=0.0, solve_mode=_lib.SOLVE_DUAL) as core:
        core.set_factor_rows(pkg.SIDE_X, 50)
        core.set_factor_rows(pkg.SIDE_Y, 40)
        core.set_matrix(pkg.SIDE_X, *csr)
        core.set_factors(pkg.SIDE_Y, M)
        core.reset_stats()
        with pytest.raises(pkg.SingularSystem):
            core.half_iteration(pkg.SIDE_X)
        assert core.stats()["rows_dual"] == 0
    with pytest.raises(oracle.SingularMatrix):
        oracle.half_iteration(*csr, M, lam=0.0, threads=2)


@pytest.mark.parametrize("flags", [pkg.FLAG_RECONSTRUCT_R, pkg.FLAG_LOSS_IGNORES_UNSPECIFIED])
def test_other_modes_stay_direct(flags):
    k = 64
    lengths = np.random.default_rng(8).integers(1, 33, size=100)
    csr, M = rows_problem(lengths, 500, k, seed=14)
    X, st = solve_x(k, csr, M, flags=flags, solve_mode=_lib.SOLVE_DUAL)
    Xo = oracle.half_iteration(*csr, M, flags=flags, threads=2)
    assert st["rows_dual"] == 0
    assert rel(X, Xo) < REL_TOL


def test_full_iterations_and_chunked_solves_with_dual_rows():
    """Two full iterations (both sides, Gramian versions changing) and the chunked entry point."""
    from myrrix_recommender_amd import synth
    k, n_users, n_items = 64, 3000, 800
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 40000, k, seed=99, negatives=0.1)
    res = {}
    for name, kw in (("dual", dict(solve_mode=_lib.SOLVE_DUAL)), ("chunked", dict(solve_mode=_lib.SOLVE_DUAL, chunk_rows=700))):
        with pkg.ALSCore(k, **kw) as core:
            core.set_factor_rows(pkg.SIDE_X, n_users)
            core.set_factor_rows(pkg.SIDE_Y, n_items)
            core.set_matrix(pkg.SIDE_X, *r_csr)
            core.set_matrix(pkg.SIDE_Y, *c_csr)
            core.set_factors(pkg.SIDE_Y, Y0)
            core.reset_stats()
            for _ in range(2):
                for side in (pkg.SIDE_X, pkg.SIDE_Y):
                    core.gramian(1 - side)
                    for c in range(core.num_chunks(side)):
                        core.solve_chunk(side, c)
                    core.check()
            assert core.stats()["rows_dual"] > 0
            res[name] = (core.get_factors(pkg.SIDE_X), core.get_factors(pkg.SIDE_Y))
    Xo, Yo = None, Y0
    for _ in range(2):
        Xo = oracle.half_iteration(*r_csr, Yo, threads=4)
        Yo = oracle.half_iteration(*c_csr, Xo, threads=4)
    for name, (X, Y) in res.items():
        assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL, (name, rel(X, Xo), rel(Y, Yo))
    assert np.array_equal(res["dual"][0], res["chunked"][0]) and np.array_equal(res["dual"][1], res["chunked"][1])


def test_deterministic():
    k = 128
    lengths = np.random.default_rng(9).integers(0, 70, size=500)
    csr, M = rows_problem(lengths, 1000, k, seed=15)
    a, _ = solve_x(k, csr, M)
    b, _ = solve_x(k, csr, M)
    assert np.array_equal(a, b)


def test_column_index_outside_the_replica_is_rejected():
    k = 16
    row_ptr = np.array([0, 2, 3], dtype=np.int64)
    val = np.ones(3, dtype=np.float32)
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, 2)
        core.set_factor_rows(pkg.SIDE_Y, 5)
        with pytest.raises(pkg.MalsError) as e:
            core.set_matrix(pkg.SIDE_X, row_ptr, np.array([0, 5, 1], dtype=np.int32), val)
        assert e.value.status == _lib.INVALID_ARG
        with pytest.raises(pkg.MalsError) as e:
            core.set_matrix(pkg.SIDE_X, row_ptr, np.array([0, -1, 1], dtype=np.int32), val)
        assert e.value.status == _lib.INVALID_ARG
        core.set_matrix(pkg.SIDE_X, row_ptr, np.array([0, 4, 1], dtype=np.int32), val)
    # the replica declared after the matrix: caught at solve time
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, 2)
        core.set_matrix(pkg.SIDE_X, row_ptr, np.array([0, 7, 1], dtype=np.int32), val)
        core.set_factor_rows(pkg.SIDE_Y, 5)
        core.set_factors(pkg.SIDE_Y, np.ones((5, k), dtype=np.float32))
        core.gramian(pkg.SIDE_Y)
        with pytest.raises(pkg.MalsError) as e:
            core.solve_side(pkg.SIDE_X)
        assert e.value.status == _lib.INVALID_ARG


def test_auto_mode_skips_the_rotation_when_few_rows_would_use_it():
    """AUTO takes the dual path only when it pays: rotating the gathered matrix costs about 1/30 of what one
    dual row saves, so 100 short rows against 50 000 factor rows stay on the direct kernels (same results)."""
    k = 64
    lengths = np.random.default_rng(10).integers(1, 33, size=100)
    csr, M = rows_problem(lengths, 50000, k, seed=16)
    Xa, sta = solve_x(k, csr, M)
    Xd, std = solve_x(k, csr, M, solve_mode=_lib.SOLVE_DUAL)
    assert sta["rows_dual"] == 0 and sta["rotate_launches"] == 0
    assert std["rows_dual"] == 100
    Xo = oracle.half_iteration(*csr, M, threads=4)
    assert rel(Xa, Xo) < REL_TOL and rel(Xd, Xo) < REL_TOL


@pytest.mark.parametrize("k", [50, 64, 128])
def test_three_stream_overlap_gives_the_same_bits(k):
    """MALS_OVERLAP=1 (off by default) enqueues a chunk's rows list, long rows and dual lists on
    three streams between a fork and a join event: the kernels and their inputs are the same, so are the factors -- bit
    for bit -- over two chained half-iterations with chunking, long rows (segments + finish) and dual rows."""
    import os
    from myrrix_recommender_amd import synth
    n_users, n_items = 3000, 900
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 60000, k, seed=77, negatives=0.05)
    res = []
    for overlap in ("0", "1"):
        os.environ["MALS_OVERLAP"] = overlap
        try:
            with pkg.ALSCore(k, segment_nnz=64, chunk_rows=700) as core:
                core.set_factor_rows(pkg.SIDE_X, n_users)
                core.set_factor_rows(pkg.SIDE_Y, n_items)
                core.set_matrix(pkg.SIDE_X, *r_csr)
                core.set_matrix(pkg.SIDE_Y, *c_csr)
                core.set_factors(pkg.SIDE_Y, Y0)
                core.reset_stats()
                core.half_iteration(pkg.SIDE_X)
                core.half_iteration(pkg.SIDE_Y)
                st = core.stats()
                res.append((core.get_factors(pkg.SIDE_X), core.get_factors(pkg.SIDE_Y)))
        finally:
            del os.environ["MALS_OVERLAP"]
        assert st["rows_solved"] == n_users + n_items and st["rows_dual"] > 0
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    Xo = oracle.half_iteration(*r_csr, Y0, threads=4)
    Yo = oracle.half_iteration(*c_csr, Xo, threads=4)
    assert rel(res[1][0], Xo) < 1e-4 and rel(res[1][1], Yo) < 1e-4
