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
    # the outlier's own row: cond(W) ~ 1e4, 1e-3 off in fp32 alone; marked and refined (als_refine_kernel)
    assert per_row[long_row] < REL_TOL, per_row[long_row]
    assert st["rows_refined"] >= 1


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


# ---- ill-conditioned rows: fp32 factorization + fp64 residuals (als_refine_kernel) ----------------------------------
def heavy_weights_problem(k, n_users, n_items, seed, flags=0):
    """alpha = 40 on values up to 150 (confidence up to 6001) against lambda = 0.01, few users: after the user
    half-iteration the item systems W = X^T X + sum 6000 x x^T + 0.4 n I have cond(W) of 1e4 .. 1e5."""
    rng = np.random.default_rng(seed)
    lengths = np.concatenate([rng.integers(1, k, size=n_users // 2), rng.integers(k, 3 * k, size=n_users - n_users // 2)])
    lengths = np.minimum(lengths, n_items)
    (row_ptr, col, val), Y0 = rows_problem(lengths, n_items, k, seed=seed + 1)
    val = (val * np.float32(30.0)).astype(np.float32)
    rows = np.repeat(np.arange(n_users), lengths)
    if flags & 2:   # lossIgnoresUnspecified: an item nobody touched would be singular in the reference too
        missing = np.setdiff1d(np.arange(n_items), col)
        rows = np.concatenate([rows, rng.integers(0, n_users, size=len(missing))])
        col = np.concatenate([col, missing.astype(np.int32)])
        val = np.concatenate([val, np.full(len(missing), 30.0, dtype=np.float32)])

    def csr(r, c, v, n):
        order = np.lexsort((c, r))
        ptr = np.zeros(n + 1, dtype=np.int64)
        np.add.at(ptr, r.astype(np.int64) + 1, 1)
        return np.cumsum(ptr).astype(np.int64), c[order].astype(np.int32), v[order].astype(np.float32)

    return csr(rows, col, val, n_users), csr(col, rows, val, n_items), (Y0 * np.float32(0.2)).astype(np.float32)


@pytest.mark.parametrize("k,flags,segment_nnz,solve_mode", [(49, 0, 0, 0), (64, 0, 0, _lib.SOLVE_DIRECT), (64, 0, 0, _lib.SOLVE_DUAL),
                                                             (127, 0, 0, 0), (30, 0, 0, 0), (128, 0, 32, 0), (100, 1, 0, 0), (80, 3, 0, 0)])
def test_heavy_confidence_weights_are_refined(k, flags, segment_nnz, solve_mode):
    """The fp32 path alone is 1e-4 .. 1e-3 off the fp64 reference on such systems.  The solving kernels mark the
    rows (largest entry of W over smallest pivot) and als_refine_kernel brings them to the reference -- short rows
    (dual path), medium rows, and rows longer than segment_nnz (segment + finish kernels) alike."""
    n_users, n_items = 120, 430
    r_csr, c_csr, Y0 = heavy_weights_problem(k, n_users, n_items, 7000 + k, flags)
    kw = dict(alpha=40.0, lam=0.01, flags=flags)
    Xo = oracle.half_iteration(*r_csr, Y0, threads=4, **kw)
    Yo = oracle.half_iteration(*c_csr, Xo, threads=4, **kw)
    res = {}
    for limit in (None, 0.0):
        with pkg.ALSCore(k, segment_nnz=segment_nnz, solve_mode=solve_mode, **kw) as core:
            if limit is not None:
                core.set_refine_limit(limit)
            core.set_factor_rows(pkg.SIDE_X, n_users)
            core.set_factor_rows(pkg.SIDE_Y, n_items)
            core.set_matrix(pkg.SIDE_Y, *c_csr)
            core.set_factors(pkg.SIDE_X, Xo)
            core.reset_stats()
            core.half_iteration(pkg.SIDE_Y)
            core.check()
            res[limit] = (core.get_factors(pkg.SIDE_Y), core.stats())
    (Y, st), (Y_off, st_off) = res[None], res[0.0]
    per_row = np.linalg.norm(Y - Yo, axis=1) / max(np.linalg.norm(Yo) / np.sqrt(len(Yo)), 1e-30)
    assert st_off["rows_refined"] == 0
    if not (flags & 1):   # reconstructR has no confidence weights: nothing to mark
        assert st["rows_refined"] > n_items // 2, st["rows_refined"]
    assert rel(Y, Yo) < 1e-5 and per_row.max() < REL_TOL, (k, flags, rel(Y, Yo), per_row.max(), st["rows_refined"])
    assert rel(Y, Yo) <= rel(Y_off, Yo) * 1.05


@pytest.mark.parametrize("seed,solve_mode,gramian_mode,alone_wrong", [
    (103, _lib.SOLVE_DIRECT, _lib.GRAMIAN_FP32, True),
    # seed 103 draws chunk_rows = 97: until round 4 only the FIRST chunk of a multi-chunk half-iteration took the dual path
    #  and the direct fp32 kernel missed the bar on the others' short rows; with the dual path in every chunk the
    # short rows -- S = I + Z Z^T is well conditioned where W is not -- are within 1e-6 without any refinement
    (103, _lib.SOLVE_DUAL, 0, False),
    (103, 0, 0, False),   # (AUTO = dual at k = 49: the same 436 short rows in every chunk)
    (175, 0, 0, True), (175, _lib.SOLVE_DUAL, _lib.GRAMIAN_FP32, True), (176, 0, 0, True)])
def test_systems_the_fp32_path_alone_gets_wrong(seed, solve_mode, gramian_mode, alone_wrong):
    """Cases of the seeded sweep (tests/test_gpu_fuzz.py) on which the fp32 accumulation + factorization alone misses
    the 1e-4 bar by up to 8x (cond(W) 3e3 .. 7e4), in every arithmetic and solve mode: with the marks and
    als_refine_kernel the same kernels land within 1e-6 of the reference."""
    from tests.test_gpu_fuzz import draw_case
    k, n_users, n_items, n_stale, r_csr, c_csr, Y0, cfg = draw_case(seed)
    cfg = dict(cfg, solve_mode=solve_mode, gramian_mode=gramian_mode)
    kw = dict(alpha=cfg["alpha"], lam=cfg["lam"], flags=cfg["flags"], threads=4)
    Xo = oracle.half_iteration(*r_csr, Y0, **kw)
    Yo = oracle.half_iteration(*c_csr, Xo, **kw)
    err = {}
    for limit in (None, 0.0):
        with pkg.ALSCore(k, **cfg) as core:
            if limit is not None:
                core.set_refine_limit(limit)
            core.set_factor_rows(pkg.SIDE_X, n_users)
            core.set_factor_rows(pkg.SIDE_Y, n_items + n_stale)
            core.set_matrix(pkg.SIDE_Y, *c_csr)
            core.set_factors(pkg.SIDE_X, Xo)
            core.set_factors(pkg.SIDE_Y, Y0)
            core.reset_stats()
            core.half_iteration(pkg.SIDE_Y)
            core.check()
            err[limit] = (rel(core.get_factors(pkg.SIDE_Y)[:n_items], Yo), core.stats()["rows_refined"])
    assert err[0.0][1] == 0, err
    assert (err[0.0][0] > 8e-5) if alone_wrong else (err[0.0][0] < 1e-5), err
    assert err[None][0] < 1e-5 and (err[None][1] > 0 or not alone_wrong), err


# 2145, 2550: reconstructR on a Gramian of fewer factor rows than features -- the reference's M^T M rounds every product
# to fp32 (MU:232), which moves its answer by 8e-5 / 2.5e-4 there; gramian_ref_kernel forms that matrix for the marked rows
# 61320, 69737: default mode, item rows holding half of all users: ratio between 64 and 128, 1.6e-4 under a limit of 128
@pytest.mark.parametrize("seed,bar", [(573, 1e-5), (1085, 1e-5), (1492, 1e-5), (2145, 1e-5), (2550, 1e-5), (61320, 1e-5), (69737, 1e-5)])
def test_rows_without_a_usable_fp32_factor_take_the_fp64_restatement(seed, bar):
    """Sweep cases (MALS_FUZZ_SEEDS=3000) with cond(W) of 1e7 and more: lossIgnoresUnspecified / reconstructR with
    lambda = 0.01 and factor rows of norm 30, or a Gramian of fewer factor rows than features.  fp32 pivots are noise
    there -- the kernels used to call such rows singular (573, 1085) or return them 4e-4 .. 2e-2 off (1492, 2145) --
    while the reference solves them (its own answer is 1e-2 away from exact arithmetic on 1085: parity is with ITS
    arithmetic).  als_exact_kernel restates that arithmetic in fp64, fp32-rounded products included, and
    als_refine_kernel (default mode) runs CG on the exact system."""
    from tests.test_gpu_fuzz import draw_case
    k, n_users, n_items, n_stale, r_csr, c_csr, Y0, cfg = draw_case(seed)
    kw = dict(alpha=cfg["alpha"], lam=cfg["lam"], flags=cfg["flags"], threads=4)
    Xo = oracle.half_iteration(*r_csr, Y0, **kw)
    Yo = oracle.half_iteration(*c_csr, Xo, **kw)
    with pkg.ALSCore(k, **cfg) as core:
        core.set_factor_rows(pkg.SIDE_X, n_users)
        core.set_factor_rows(pkg.SIDE_Y, n_items + n_stale)
        core.set_matrix(pkg.SIDE_Y, *c_csr)
        core.set_factors(pkg.SIDE_X, Xo)
        core.set_factors(pkg.SIDE_Y, Y0)
        core.reset_stats()
        core.half_iteration(pkg.SIDE_Y)
        core.check()                      # no SingularSystem: the reference does not throw either
        Y = core.get_factors(pkg.SIDE_Y)[:n_items]
        st = core.stats()
    assert st["rows_refined"] > 0
    assert rel(Y, Yo) < bar, (seed, rel(Y, Yo), st["rows_refined"])


def test_refinement_leaves_well_conditioned_problems_alone():
    """The reference's default hyper-parameters on ordinary data: nothing is marked, and the factors are bit for bit
    what they are with the refinement switched off."""
    k = 64
    csr, M = rows_problem(mixed_lengths(k, 800, 11), 2000, k, seed=31)
    X, st = solve_x(k, csr, M)
    X0, _ = solve_x(k, csr, M, refine_limit=0.0)
    assert st["rows_refined"] == 0
    assert np.array_equal(X, X0)


def test_operand_range_check_switches_to_the_fp32_gather():
    """One value 3e3 times the others AND one factor element 2e4 times the others over a 200K-row factor matrix stretch
    bound / typical operand (sqrt(w_max / w_mean) * max |y| / rms |y| ~ 2^18) past the 16 binades both f16 halves can
    hold: the launch must run the fp32-gather kernels (bitwise the GRAMIAN_FP32 result), with no host
    round trip (the split kernels return at once on the device-side flag).  (Until round 3 the bound on |y| was
    sqrt(max_f G_ff) and one value of 1e7 alone tripped the flag; with the exact maximum a single outlier VALUE cannot --
    it drags the mean weight along --, see test_operand_bound_is_the_exact_maximum_not_the_gramian_diagonal.)"""
    k = 64
    lengths = np.random.default_rng(4).integers(40, 300, size=500)
    csr, M = rows_problem(lengths, 200000, k, seed=24)
    v = csr[2].copy()
    v[7] = 1.0e4
    csr = (csr[0], csr[1], v)
    M[123, 5] *= 2.0e4
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


@pytest.mark.parametrize("lds_gather", ["1", "0"])
def test_forced_range_flag_at_k128_is_the_fp32_gather_bit_for_bit(lds_gather):
    """The same switch at k = 128, where the split-precision kernels are the LDS-staged ones (csrc/lds_kernels.h, feature order
    permuted, their own Gramian image and finish kernels): with the device-side flag down every one of them returns at once
    and the fp32 kernels queued behind them produce the GRAMIAN_FP32 result bit for bit -- fused rows, segmented rows
    (three rows long enough for the segments + finish path at this segment size) and both settings of MALS_LDS_GATHER."""
    k = 128
    rng = np.random.default_rng(6)
    lengths = np.concatenate([rng.integers(65, 400, size=300), [3000, 5000, 9000]])
    csr, M = rows_problem(lengths, 20000, k, seed=27)
    kw = dict(solve_mode=_lib.SOLVE_DIRECT, segment_nnz=1024)
    old = os.environ.get("MALS_LDS_GATHER")
    os.environ["MALS_LDS_GATHER"] = lds_gather
    try:
        B, _ = solve_x(k, csr, M, gramian_mode=_lib.GRAMIAN_FP32, **kw)
        A, _ = solve_x(k, csr, M, **kw)
        assert not np.array_equal(A, B) and rel(A, B) < 1e-5, rel(A, B)     # ordinary data: the split kernels ran
        Xo = oracle.half_iteration(*csr, M, threads=4)
        assert rel(A, Xo) < 1e-5 and rel(B, Xo) < 1e-5, (rel(A, Xo), rel(B, Xo))
        os.environ["MALS_FORCE_RANGE_FLAG"] = "0"
        try:
            C, _ = solve_x(k, csr, M, **kw)
        finally:
            del os.environ["MALS_FORCE_RANGE_FLAG"]
        assert np.array_equal(C, B)
    finally:
        if old is None:
            del os.environ["MALS_LDS_GATHER"]
        else:
            os.environ["MALS_LDS_GATHER"] = old


def test_negative_alpha_runs_the_fp32_gather():
    """alpha < 0 (accepted by the reference, ALS:506-509) has no real sqrt(alpha |r|)."""
    k = 64
    csr, M = rows_problem(np.full(50, 40), 500, k, seed=25, negatives=0.0)
    X, _ = solve_x(k, csr, M, alpha=-0.001)
    Xo = oracle.half_iteration(*csr, M, alpha=-0.001, threads=2)
    assert rel(X, Xo) < REL_TOL


def test_operand_bound_is_the_exact_maximum_not_the_gramian_diagonal():
    """The split-precision gather scales its operands with a bound on |y|.  sqrt(max_f G_ff) is loose by sqrt(rows) --
    10 binades at 1e6 rows, 13 at the 1e8 rows of C5's X -- and with heavy-tailed values (one play count of 20 000 among
    1..5) that alone pushed the range flag over its 16 binades: the whole half-iteration fell back to the fp32 gather.
    The Gramian kernels now record the exact max |element| (round 3): same data, split-precision kernels, same answer."""
    k, n_items = 64, 1_200_000
    lengths = np.random.default_rng(5).integers(49, 200, size=1500)        # direct rows (k = 64: dual up to 48 entries)
    csr, M = rows_problem(lengths, n_items, k, seed=26, negatives=0.0)
    v = csr[2].copy()
    v[csr[0][3] + 1] = 2.0e4
    csr = (csr[0], csr[1], v)
    n_rows = len(lengths)
    G = oracle.gramian(M)
    Xo = oracle.solve_rows(*csr, M, G, threads=8)
    with pkg.ALSCore(k, solve_mode=_lib.SOLVE_DIRECT) as core:
        core.set_factor_rows(pkg.SIDE_X, n_rows)
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_matrix(pkg.SIDE_X, *csr)
        core.set_factors(pkg.SIDE_Y, M)
        core.half_iteration(pkg.SIDE_X)              # G by this library's kernels: exact bound
        S, _, flag, bound = core.gather_scale()
        X = core.get_factors(pkg.SIDE_X)
        assert flag == 1.0, "the split-precision kernels must run"
        assert abs(bound - float(np.abs(M).max())) <= 1e-6 * bound
        assert rel(X, Xo) < REL_TOL, rel(X, Xo)
        # the same Gramian handed in from outside: no maximum came with it -> the diagonal's bound, which trips the flag
        core.set_gramian(pkg.SIDE_Y, G)
        core.solve_side(pkg.SIDE_X)
        core.check()
        S2, _, flag2, bound2 = core.gather_scale()
        X2 = core.get_factors(pkg.SIDE_X)
        assert bound2 > 100.0 * bound and S2 < S and flag2 == 0.0
        assert rel(X2, Xo) < REL_TOL, rel(X2, Xo)     # the fp32 twins: correct as well, just slower
