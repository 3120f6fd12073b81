"""top-N scoring on the device (mals_recommend / mals_recommend_vectors)
against the oracle's restatement of RecommendIterator + TopN."""
import os

import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import synth
from oracle import topn_oracle as to

pytestmark = pytest.mark.gpu


def make(k, n_users, n_items, nnz, seed):
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, nnz, k, seed=seed)
    core = pkg.ALSCore(k)
    core.set_factor_rows(pkg.SIDE_X, n_users)
    core.set_factor_rows(pkg.SIDE_Y, n_items)
    core.set_matrix(pkg.SIDE_X, *r_csr)
    core.set_matrix(pkg.SIDE_Y, *c_csr)
    core.set_factors(pkg.SIDE_Y, Y0)
    core.half_iteration(pkg.SIDE_X)
    core.half_iteration(pkg.SIDE_Y)
    return core, r_csr, core.get_factors(pkg.SIDE_X), core.get_factors(pkg.SIDE_Y)


def same_ranking(idx, sc, oidx, osc, atol=None):
    """The scores ARE the reference's (every product rounded to fp32, fp64 sum in feature order, one cast: the device
    computes them with exactly those operations) and ties go by ascending index on both sides: indices and score bits
    are identical, nothing is tolerated."""
    n = len(oidx)
    assert np.all(idx[n:] == -1)
    assert np.array_equal(idx[:n], oidx), (idx[:n], oidx)
    assert np.array_equal(sc[:n].view(np.uint32), np.asarray(osc, np.float32).view(np.uint32)), (sc[:n], osc)


@pytest.mark.parametrize("k", [2, 10, 30, 64, 100])
def test_recommend_matches_oracle(k):
    core, r_csr, X, Y = make(k, 300, 1000, 20000, 70 + k)
    with core:
        users = np.array([0, 5, 17, 123, 299], np.int64)
        for consider_known in (False, True):
            idx, sc, cnt = core.recommend(users, 10, consider_known_items=consider_known)
            for q, u in enumerate(users):
                known = None if consider_known else r_csr[1][r_csr[0][u]:r_csr[0][u + 1]]
                oidx, osc = to.recommend(Y, X[u], 10, known)
                assert cnt[q] == len(oidx)
                same_ranking(idx[q], sc[q], oidx, osc)
                if known is not None:
                    assert not set(idx[q, :cnt[q]].tolist()) & set(known.tolist())


def test_batches_larger_than_one_pass_and_many_results():
    core, r_csr, X, Y = make(16, 200, 5000, 15000, 5)
    with core:
        users = np.arange(150, dtype=np.int64)                        # 3 passes of 64 queries
        idx, sc, cnt = core.recommend(users, 300)
        for q in (0, 63, 64, 127, 128, 149):
            known = r_csr[1][r_csr[0][q]:r_csr[0][q + 1]]
            oidx, osc = to.recommend(Y, X[q], 300, known)
            same_ranking(idx[q], sc[q], oidx, osc)


def test_ties_and_short_candidate_lists():
    k = 8
    n_items = 700
    Y = np.zeros((n_items, k), np.float32)
    Y[:, 0] = np.repeat(np.arange(7, dtype=np.float32), 100)          # 100-way ties at every score
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_factors(pkg.SIDE_Y, Y)
        q = np.zeros((1, k), np.float32)
        q[0, 0] = 1.0
        idx, sc, cnt = core.recommend_vectors(q, 150)                 # 100 sixes + the 50 lowest-index fives
        oidx, osc = to.recommend(Y, q[0], 150)
        assert np.array_equal(idx[0], oidx) and np.array_equal(sc[0], osc)
        # ties beyond the selection buffer: all 700 items score 0 against a zero query
        idx, sc, cnt = core.recommend_vectors(np.zeros((1, k), np.float32), 20)
        assert idx[0].tolist() == list(range(20)) and np.all(sc[0] == 0.0)
        # fewer candidates than requested: everything but 5 items excluded
        excl = [list(range(5, n_items))]
        idx, sc, cnt = core.recommend_vectors(q, 10, exclude=excl)
        assert cnt[0] == 5 and idx[0, :5].tolist() == [0, 1, 2, 3, 4] and np.all(idx[0, 5:] == -1)


def test_argument_checks():
    with pkg.ALSCore(4) as core:
        with pytest.raises(pkg.MalsError):
            core.recommend(np.array([0], np.int64), 5)
        core.set_factor_rows(pkg.SIDE_X, 2)
        core.set_factor_rows(pkg.SIDE_Y, 3)
        with pytest.raises(pkg.MalsError):
            core.recommend(np.array([0], np.int64), 5)                # known items need the matrix
        with pytest.raises(pkg.MalsError):
            core.recommend(np.array([7], np.int64), 5, consider_known_items=True)
        idx, sc, cnt = core.recommend(np.array([1], np.int64), 5, consider_known_items=True)
        assert cnt[0] == 3                                            # three items, all scoring 0


# ---- large catalogues: the filter path (sample -> bound -> one filtered pass over Y -> exact scores of the candidates) ------
def big_core(k, n_items, n_users, deg, seed):
    rng = np.random.default_rng(seed)
    Y = (rng.standard_normal((n_items, k)) / np.sqrt(k)).astype(np.float32)
    X = (rng.standard_normal((n_users, k)) / np.sqrt(k)).astype(np.float32)
    rp = np.arange(n_users + 1, dtype=np.int64) * deg
    col = np.concatenate([np.sort(rng.choice(n_items, deg, replace=False)) for _ in range(n_users)]).astype(np.int32)
    core = pkg.ALSCore(k)
    core.set_factor_rows(pkg.SIDE_X, n_users)
    core.set_factor_rows(pkg.SIDE_Y, n_items)
    core.set_factors(pkg.SIDE_X, X)
    core.set_factors(pkg.SIDE_Y, Y)
    core.set_matrix(pkg.SIDE_X, rp, col, np.ones(len(col), np.float32))
    return core, X, Y, rp, col


@pytest.mark.parametrize("k,n_items,how_many", [(64, 140_000, 10), (16, 300_000, 50), (100, 150_000, 5), (30, 200_000, 64), (128, 131_072, 1)])
def test_filter_path_matches_oracle(k, n_items, how_many):
    core, X, Y, rp, col = big_core(k, n_items, 70, 300, 11 + k)
    with core:
        users = np.arange(70, dtype=np.int64)
        idx, sc, cnt = core.recommend(users, how_many)
        idx2, sc2, _ = core.recommend(users, how_many, consider_known_items=True)
        for q in (0, 1, 33, 63, 64, 69):
            known = col[rp[q]:rp[q + 1]]
            oidx, osc = to.recommend(Y, X[q], how_many, known)
            assert cnt[q] == how_many
            same_ranking(idx[q], sc[q], oidx, osc)
            assert not set(idx[q].tolist()) & set(known.tolist())
            oidx, osc = to.recommend(Y, X[q], how_many)
            same_ranking(idx2[q], sc2[q], oidx, osc)


@pytest.mark.parametrize("k,n_queries", [(64, 250), (80, 170), (16, 700), (128, 120), (50, 190)])
def test_filter_path_full_passes_every_tile_count_and_all_slots(k, n_queries):
    """Passes with 1-4 query tiles per wave (240 / 160 / 112 queries per pass at 32-64 / 65-96 / 97-128 features), a ragged last
    pass, and more passes than streams: every query of the batch answered as if it were alone."""
    core, X, Y, rp, col = big_core(k, 140_000, n_queries, 40, 500 + k)
    with core:
        users = np.arange(n_queries, dtype=np.int64)
        idx, sc, cnt = core.recommend(users, 10)
        rng = np.random.default_rng(k)
        for q in sorted(set([0, 15, 16, n_queries - 1] + rng.integers(0, n_queries, 8).tolist())):
            known = col[rp[q]:rp[q + 1]]
            oidx, osc = to.recommend(Y, X[q], 10, known)
            assert cnt[q] == 10
            same_ranking(idx[q], sc[q], oidx, osc)
        one = core.recommend(users[37:38], 10)
        assert np.array_equal(one[0][0], idx[37]) and np.array_equal(one[1][0], sc[37])


def test_filter_path_when_the_sample_is_won_by_known_items():
    """A user whose known items are exactly the best-scoring items: the buckets they win are dropped from the threshold's
    sample, the filter still finds the best of the rest."""
    k, n_items = 32, 150_000
    rng = np.random.default_rng(99)
    Y = (rng.standard_normal((n_items, k)) / np.sqrt(k)).astype(np.float32)
    X = (rng.standard_normal((3, k)) / np.sqrt(k)).astype(np.float32)
    order = np.argsort(-(Y.astype(np.float64) @ X[1].astype(np.float64)))
    known1 = np.sort(order[:5000]).astype(np.int32)                    # user 1 knows its 5000 best items
    rp = np.array([0, 3, 3 + len(known1), 3 + len(known1)], dtype=np.int64)
    col = np.concatenate([np.array([1, 2, 3], np.int32), known1])
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, 3)
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_factors(pkg.SIDE_X, X)
        core.set_factors(pkg.SIDE_Y, Y)
        core.set_matrix(pkg.SIDE_X, rp, col, np.ones(len(col), np.float32))
        idx, sc, cnt = core.recommend(np.arange(3, dtype=np.int64), 20)
        for q in range(3):
            known = col[rp[q]:rp[q + 1]]
            oidx, osc = to.recommend(Y, X[q], 20, known)
            same_ranking(idx[q], sc[q], oidx, osc)
            assert not set(idx[q].tolist()) & set(known.tolist())


def test_queries_that_overflow_their_candidate_buffer_fall_back_one_by_one(monkeypatch):
    """A sample far too small for the catalogue (tuning override): the thresholds are low, some queries of the pass collect more
    candidates than their buffer holds and are answered by the dense path, the others by the filter -- all of them exactly."""
    core, X, Y, rp, col = big_core(32, 200_000, 48, 30, 4242)
    with core:
        users = np.arange(48, dtype=np.int64)
        want = core.recommend(users, 10)
        monkeypatch.setenv("MALS_TOPN_SAMPLE_ITEMS", "1100")
        got = core.recommend(users, 10)
        assert all(np.array_equal(a, b) for a, b in zip(want, got))
        for q in (0, 7, 47):
            oidx, osc = to.recommend(Y, X[q], 10, col[rp[q]:rp[q + 1]])
            same_ranking(got[0][q], got[1][q], oidx, osc)


def test_recommend_sees_the_half_iteration_enqueued_before_it():
    """The passes of a call run on the library's own streams: they must wait for what is already on the handle's stream.  A
    half-iteration (asynchronous) rewrites X; the recommendations that follow are those of the NEW user vectors."""
    k, n_items, n_users = 32, 140_000, 300
    rng = np.random.default_rng(31)
    Y = (rng.standard_normal((n_items, k)) / np.sqrt(k)).astype(np.float32)
    X0 = np.zeros((n_users, k), np.float32)                             # stale user vectors: all-zero scores
    deg = 20
    rp = np.arange(n_users + 1, dtype=np.int64) * deg
    col = np.concatenate([np.sort(rng.choice(n_items, deg, replace=False)) for _ in range(n_users)]).astype(np.int32)
    val = rng.uniform(0.5, 3.0, len(col)).astype(np.float32)
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, n_users)
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_factors(pkg.SIDE_X, X0)
        core.set_factors(pkg.SIDE_Y, Y)
        core.set_matrix(pkg.SIDE_X, rp, col, val)
        users = np.arange(n_users, dtype=np.int64)
        core.half_iteration(pkg.SIDE_X)                                 # enqueued, not waited for
        idx, sc, cnt = core.recommend(users, 10)
        X1 = core.get_factors(pkg.SIDE_X)
        assert np.abs(X1).max() > 0
        for q in (0, 17, 299):
            oidx, osc = to.recommend(Y, X1[q], 10, col[rp[q]:rp[q + 1]])
            same_ranking(idx[q], sc[q], oidx, osc)


def test_a_million_items_at_the_bench_shape():
    """The shape tools/bench_topn.py times (1M items, k = 64, N = 10, known items skipped), a 4096-query call: sampled queries
    against the oracle, and every query's list a strictly descending-or-tied sequence of finite scores with no known item."""
    k, n_items, n_users, deg = 64, 1_000_000, 5000, 100
    rng = np.random.default_rng(20260929)
    Y = (rng.standard_normal((n_items, k)) / np.sqrt(k)).astype(np.float32)
    X = (rng.standard_normal((n_users, k)) / np.sqrt(k)).astype(np.float32)
    rp = np.arange(n_users + 1, dtype=np.int64) * deg
    col = np.sort(rng.integers(0, n_items, (n_users, deg)), axis=1).astype(np.int32).ravel()
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, n_users)
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_factors(pkg.SIDE_X, X)
        core.set_factors(pkg.SIDE_Y, Y)
        core.set_matrix(pkg.SIDE_X, rp, col, np.ones(len(col), np.float32))
        users = rng.integers(0, n_users, 4096).astype(np.int64)
        idx, sc, cnt = core.recommend(users, 10)
        assert (cnt == 10).all() and np.isfinite(sc).all() and (np.diff(sc, axis=1) <= 0).all()
        known = col.reshape(n_users, deg)[users]
        assert not (idx[:, :, None] == known[:, None, :]).any()
        for q in (0, 239, 240, 4095, int(rng.integers(0, 4096))):
            u = users[q]
            oidx, osc = to.recommend(Y, X[u], 10, col[rp[u]:rp[u + 1]])
            same_ranking(idx[q], sc[q], oidx, osc)


def test_filter_path_equals_full_path(monkeypatch):
    core, X, Y, rp, col = big_core(32, 160_000, 20, 500, 3)
    with core:
        users = np.arange(20, dtype=np.int64)
        a = core.recommend(users, 25)
        monkeypatch.setenv("MALS_TOPN_FULL", "1")
        b = core.recommend(users, 25)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_filter_path_with_massive_ties_and_exclusions():
    k, n_items = 8, 200_000
    Y = np.zeros((n_items, k), np.float32)
    Y[:, 0] = (np.arange(n_items) % 7).astype(np.float32)             # 7 distinct scores, ~28K items each
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_factors(pkg.SIDE_Y, Y)
        q = np.zeros((2, k), np.float32)
        q[:, 0] = 1.0
        excl = [[6, 13, 20], []]                                       # the three lowest-index items scoring 6
        idx, sc, cnt = core.recommend_vectors(q, 12, exclude=excl)
        for j in range(2):
            oidx, osc = to.recommend(Y, q[j], 12, excl[j])
            assert np.array_equal(idx[j], oidx) and np.array_equal(sc[j], osc)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MALS_TOPN_SEEDS", "40"))))   # MALS_TOPN_SEEDS=400: a longer hunt
def test_seeded_recommend_sweep(seed):
    """Random feature counts, catalogue sizes (below and above the filter path's threshold), result counts, query
    batches, known-item handling and score structures (ties, negative scores, fewer candidates than results) against
    the oracle's RecommendIterator + TopN."""
    rng = np.random.default_rng(77_000 + seed)
    k = int(rng.choice([1, 2, 7, 16, 30, 33, 64, 100, 128]))
    n_items = int(rng.choice([1, 5, 63, 64, 65, 1000, 4097, 30000, 70000, 140000]))
    n_users = int(rng.integers(1, 200))
    how_many = int(rng.choice([1, 2, 10, 64, 300]))
    X = rng.standard_normal((n_users, k)).astype(np.float32)
    Y = rng.standard_normal((n_items, k)).astype(np.float32)
    mode = int(rng.integers(0, 4))
    if mode == 1:      # massive ties: a handful of distinct item vectors
        Y = Y[rng.integers(0, min(n_items, 4), size=n_items)]
    elif mode == 2:    # all scores negative for half of the users
        Y = np.abs(Y)
        X[::2] = -np.abs(X[::2])
    elif mode == 3:    # quantised scores
        X, Y = np.round(X), np.round(Y)
    lens = np.minimum(rng.integers(0, max(2, min(n_items, 40)), size=n_users), n_items)
    if rng.random() < 0.3:
        lens[rng.integers(0, n_users)] = n_items          # a user who knows every item
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    col = np.concatenate([np.sort(rng.choice(n_items, size=int(n), replace=False)) for n in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    val = np.ones(len(col), dtype=np.float32)
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, n_users)
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_matrix(pkg.SIDE_X, row_ptr, col, val)
        core.set_factors(pkg.SIDE_X, X)
        core.set_factors(pkg.SIDE_Y, Y)
        users = np.sort(rng.choice(n_users, size=min(n_users, int(rng.choice([1, 3, 64, 65, 130]))), replace=False)).astype(np.int64)
        for consider_known in (False, True):
            idx, sc, cnt = core.recommend(users, how_many, consider_known_items=consider_known)
            for q, u in enumerate(users):
                known = None if consider_known else col[row_ptr[u]:row_ptr[u + 1]]
                oidx, osc = to.recommend(Y, X[u], how_many, known)
                assert cnt[q] == len(oidx), (seed, q, cnt[q], len(oidx))
                same_ranking(idx[q], sc[q], oidx, osc)


# ---- queries of several vectors: recommendToMany (ServerRecommender.java:366-441, RecommendIterator.java:93-104) -------
@pytest.mark.parametrize("k,n_items,how_many", [(10, 3000, 7), (64, 140_000, 10), (33, 200_000, 20)])
def test_recommend_to_many_matches_oracle(k, n_items, how_many):
    rng = np.random.default_rng(500 + k)
    Y = (rng.standard_normal((n_items, k)) / np.sqrt(k)).astype(np.float32)
    sizes = [1, 2, 3, 7, 1, 5, 40, 2] + [int(rng.integers(1, 6)) for _ in range(300)]
    queries = [rng.standard_normal((n, k)).astype(np.float32) for n in sizes]
    excl = [np.sort(rng.choice(n_items, int(rng.integers(0, 30)), replace=False)) for _ in sizes]
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_factors(pkg.SIDE_Y, Y)
        idx, sc, cnt = core.recommend_to_many(queries, how_many, exclude=excl)
        for q in list(range(8)) + [63, 64, 255, 256, 257, 307]:
            oidx, osc = to.recommend(Y, queries[q], how_many, excl[q])
            assert cnt[q] == how_many
            same_ranking(idx[q], sc[q], oidx, osc)
        # the mean of the dots is not the dot with the mean: a single-vector query built from the mean ranks differently somewhere
        with pytest.raises(pkg.MalsError):
            core.recommend_to_many([np.zeros((0, k), np.float32)], 3)            # "features must not be empty"


def test_near_ties_are_resolved_like_the_reference():
    """Items whose scores differ only in the last bits, by construction: duplicates of one vector with single-ulp
    perturbations.  The ranking among them is decided by the reference's own rounding sequence."""
    rng = np.random.default_rng(9)
    k, n_items = 64, 150_000
    base = (rng.standard_normal(k) / np.sqrt(k)).astype(np.float32)
    Y = (rng.standard_normal((n_items, k)) * 0.01).astype(np.float32)
    hot = rng.choice(n_items, 400, replace=False)
    Yh = np.tile(base, (400, 1))
    bits = Yh.view(np.uint32).copy()
    bits += rng.integers(0, 3, size=bits.shape).astype(np.uint32)     # 0..2 ulps up, per component
    Y[hot] = bits.view(np.float32)
    x = np.stack([base * 3, base * 3 * (1 + 2.0 ** -22)]).astype(np.float32)
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_factors(pkg.SIDE_Y, Y)
        idx, sc, cnt = core.recommend_vectors(x, 50)
        for q in range(2):
            oidx, osc = to.recommend(Y, x[q], 50)
            same_ranking(idx[q], sc[q], oidx, osc)
            assert len(set(sc[q].tolist())) < 50                      # real ties among the best 50
