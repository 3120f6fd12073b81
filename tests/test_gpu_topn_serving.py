"""SURVEY.md section 8(f) row 4 as the reference is ENTERED: many request threads on one generation, one user per call
(ServerRecommender.java:359-441 -> multithreadedTopN :443-508), and userTagIDs never recommended (RecommendIterator.java:72).
Every result of every thread is compared with the oracle's restatement of RecommendIterator + TopN: indices and score bits."""
import threading

import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from oracle import topn_oracle as to
from tests.test_gpu_topn import big_core, same_ranking

pytestmark = pytest.mark.gpu


def oracle_table(Y, X, users, how_many, rp, col, tags=None):
    out = {}
    for u in users:
        out[int(u)] = to.recommend(Y, X[u], how_many, col[rp[u]:rp[u + 1]], tags)
    return out


def run_threads(n_threads, fn):
    errs = []

    def wrap(t):
        try:
            fn(t)
        except BaseException as e:  # noqa: BLE001 -- reported below
            errs.append((t, e))
    th = [threading.Thread(target=wrap, args=(t,)) for t in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0][1]


@pytest.mark.parametrize("k,n_items", [(64, 200_000), (30, 140_000)])
def test_32_threads_of_single_user_calls_equal_the_oracle(k, n_items):
    """32 threads x 1000 single-user calls on one handle: every answer array_equal to the oracle's (ctypes drops the GIL
    inside a call, so the calls really are concurrent), and the calls were folded into far fewer passes than calls."""
    n_users, how_many, n_threads, per_thread = 96, 10, 32, 1000
    core, X, Y, rp, col = big_core(k, n_items, n_users, 40, 9000 + k)
    with core:
        want = oracle_table(Y, X, range(n_users), how_many, rp, col)
        before = core.recommend_front_stats()
        bad = []

        def worker(t):
            rng = np.random.default_rng(100 + t)
            for _ in range(per_thread):
                u = int(rng.integers(0, n_users))
                idx, sc, cnt = core.recommend(np.array([u], np.int64), how_many)
                oidx, osc = want[u]
                if not (cnt[0] == len(oidx) and np.array_equal(idx[0, :len(oidx)], oidx)
                        and np.array_equal(sc[0, :len(oidx)].view(np.uint32), np.asarray(osc, np.float32).view(np.uint32))):
                    bad.append((t, u))
        run_threads(n_threads, worker)
        assert not bad, bad[:5]
        st = core.recommend_front_stats()
        calls = st["calls"] - before["calls"]
        passes = st["passes"] - before["passes"]
        assert calls == n_threads * per_thread and st["queries"] - before["queries"] == calls
        assert passes < calls, (passes, calls)      # (how much they fold depends on the host; that they fold at all does not)


def test_mixed_callers_small_bulk_vectors_and_known_flags():
    """Threads of different kinds on one handle at once: single users with and without their known items, calls of a few
    users, calls that are passes of their own (300 users), caller-supplied vectors with exclusion lists, two values of
    how_many -- each answered as if it were alone."""
    k, n_items, n_users = 32, 150_000, 320
    core, X, Y, rp, col = big_core(k, n_items, n_users, 25, 777)
    with core:
        failures = []

        def check(u, hm, known, idx, sc, cnt):
            oidx, osc = to.recommend(Y, X[u], hm, col[rp[u]:rp[u + 1]] if known else None)
            if not (cnt == len(oidx) and np.array_equal(idx[:len(oidx)], oidx)
                    and np.array_equal(sc[:len(oidx)].view(np.uint32), np.asarray(osc, np.float32).view(np.uint32))):
                failures.append((u, hm, known))

        def worker(t):
            rng = np.random.default_rng(t)
            for it in range(40):
                kind = (t + it) % 5
                if kind == 0:      # one user, known items skipped
                    u = int(rng.integers(0, n_users))
                    idx, sc, cnt = core.recommend(np.array([u], np.int64), 10)
                    check(u, 10, True, idx[0], sc[0], cnt[0])
                elif kind == 1:    # one user, known items considered, another how_many
                    u = int(rng.integers(0, n_users))
                    idx, sc, cnt = core.recommend(np.array([u], np.int64), 7, consider_known_items=True)
                    check(u, 7, False, idx[0], sc[0], cnt[0])
                elif kind == 2:    # a few users
                    us = rng.integers(0, n_users, 5).astype(np.int64)
                    idx, sc, cnt = core.recommend(us, 10)
                    for q in (0, 4):
                        check(int(us[q]), 10, True, idx[q], sc[q], cnt[q])
                elif kind == 3 and it % 8 == 3:    # a call that is passes of its own
                    us = rng.integers(0, n_users, 300).astype(np.int64)
                    idx, sc, cnt = core.recommend(us, 10)
                    for q in (0, 299):
                        check(int(us[q]), 10, True, idx[q], sc[q], cnt[q])
                else:              # an anonymous user: its vector and the items it was built from
                    u = int(rng.integers(0, n_users))
                    ex = col[rp[u]:rp[u + 1]].astype(np.int64)
                    idx, sc, cnt = core.recommend_vectors(X[u:u + 1], 10, exclude=[ex])
                    check(u, 10, True, idx[0], sc[0], cnt[0])
        run_threads(12, worker)
        assert not failures, failures[:5]


def test_tag_items_are_never_recommended():
    """userTagIDs: struck for every caller, on the filter path and on the dense path, also when the tag items are exactly the
    best-scoring items of a query."""
    k, n_items, n_users = 32, 150_000, 40
    core, X, Y, rp, col = big_core(k, n_items, n_users, 30, 31337)
    with core:
        best3 = np.argsort(-(Y.astype(np.float64) @ X[3].astype(np.float64)))[:40]
        tags = np.unique(np.concatenate([best3, np.arange(1000, 1100), [0, n_items - 1]])).astype(np.int64)
        core.set_tag_items(np.concatenate([tags, [-1, n_items + 5]]))       # (entries outside Y are ignored)
        assert core.tag_item_count() == len(tags)
        users = np.arange(n_users, dtype=np.int64)
        idx, sc, cnt = core.recommend(users, 10)
        for q in range(n_users):
            oidx, osc = to.recommend(Y, X[q], 10, col[rp[q]:rp[q + 1]], tags)
            same_ranking(idx[q], sc[q], oidx, osc)
            assert not set(idx[q].tolist()) & set(tags.tolist())
        one = core.recommend(users[3:4], 10)
        assert np.array_equal(one[0][0], idx[3])
        v_idx, v_sc, _ = core.recommend_vectors(X[:5], 10)
        for q in range(5):
            oidx, osc = to.recommend(Y, X[q], 10, None, tags)
            same_ranking(v_idx[q], v_sc[q], oidx, osc)
        core.set_tag_items(None)
        assert core.tag_item_count() == 0
        idx, sc, cnt = core.recommend(users[3:4], 10)
        oidx, osc = to.recommend(Y, X[3], 10, col[rp[3]:rp[4]])
        same_ranking(idx[0], sc[0], oidx, osc)


def test_tag_items_on_the_dense_path():
    k, n_items = 10, 3000
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((n_items, k)).astype(np.float32)
    X = rng.standard_normal((4, k)).astype(np.float32)
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_factors(pkg.SIDE_Y, Y)
        tags = np.argsort(-(Y @ X[0]))[:25].astype(np.int64)
        core.set_tag_items(tags)
        idx, sc, cnt = core.recommend_vectors(X, 15)
        for q in range(4):
            oidx, osc = to.recommend(Y, X[q], 15, None, tags)
            same_ranking(idx[q], sc[q], oidx, osc)
        # a replica re-declared with another item count: the mask no longer fits, the call says so
        core.set_factor_rows(pkg.SIDE_Y, n_items + 64)
        with pytest.raises(pkg.MalsError):
            core.recommend_vectors(X, 5)


def test_known_items_of_an_old_matrix_do_not_survive_a_new_one():
    """knownItemIDs are a CSR over the local user rows; a new user-side matrix drops them (they may have been
    borrowed from an ingest that is gone, and their rows are no longer these rows)."""
    k, n_items = 8, 2000
    rng = np.random.default_rng(2)
    Y = rng.standard_normal((n_items, k)).astype(np.float32)
    X = rng.standard_normal((6, k)).astype(np.float32)
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, 6)
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_factors(pkg.SIDE_X, X)
        core.set_factors(pkg.SIDE_Y, Y)
        rp = np.arange(7, dtype=np.int64) * 2
        col = np.tile(np.array([1, 2], np.int32), 6)
        core.set_matrix(pkg.SIDE_X, rp, col, np.ones(12, np.float32))
        best = np.argsort(-(Y @ X[0]))[:3].astype(np.int32)
        kp = np.array([0, 3, 3, 3, 3, 3, 3], np.int64)
        core.set_known_items(kp, best)
        idx, _, _ = core.recommend(np.array([0], np.int64), 3)
        assert not set(idx[0].tolist()) & set(best.tolist())
        # a new matrix with fewer rows: the old known items are gone, the rows of R count again
        core.set_matrix(pkg.SIDE_X, rp[:4], col[:6], np.ones(6, np.float32))
        idx, sc, _ = core.recommend(np.array([0], np.int64), 3)
        oidx, osc = to.recommend(Y, X[0], 3, np.array([1, 2]))
        same_ranking(idx[0], sc[0], oidx, osc)


@pytest.mark.parametrize("n_items", [3000, 150_000])      # the dense path / the filter path
@pytest.mark.parametrize("bad", [np.nan, -np.nan, np.inf])
def test_a_non_finite_score_fails_the_call_like_the_reference(n_items, bad):
    """RecommendIterator.java:105: Preconditions.checkState(isFinite(result), "Bad recommendation value") -- a model with a
    non-finite factor does not produce a quietly wrong top-N (the filter's signed-integer hit test let a NaN with
    the sign bit set through unnoticed)."""
    k = 32
    rng = np.random.default_rng(8)
    Y = (rng.standard_normal((n_items, k)) / np.sqrt(k)).astype(np.float32)
    X = (rng.standard_normal((4, k)) / np.sqrt(k)).astype(np.float32)
    X[:, 3] = np.abs(X[:, 3])                                  # (+inf times these is +inf, not -inf)
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_factors(pkg.SIDE_Y, Y)
        idx, sc, cnt = core.recommend_vectors(X, 5)
        oidx, osc = to.recommend(Y, X[0], 5)
        same_ranking(idx[0], sc[0], oidx, osc)
        Y[n_items // 2, 3] = np.float32(bad)
        core.set_factors(pkg.SIDE_Y, Y)
        with pytest.raises(pkg.MalsError, match="Bad recommendation value"):
            core.recommend_vectors(X, 5)
        # the item excluded for every query: the reference never scores it (RecommendIterator.java:75-82), nothing to report
        idx, sc, cnt = core.recommend_vectors(X, 5, exclude=[[n_items // 2]] * 4)
        Yc = Y.copy()
        Yc[n_items // 2] = 0
        oidx, osc = to.recommend(Yc, X[0], 5, [n_items // 2])
        same_ranking(idx[0], sc[0], oidx, osc)


def test_anonymous_and_to_many_callers_are_folded_too():
    """recommendToAnonymous / recommendToMany from request threads (SR:366-441, 561-606): small by-vector calls -- one vector
    with the anonymous user's items excluded, or a few users' vectors as one query -- share passes like the by-user calls, each
    answered as if alone (multi-vector score = the reference's mean of dots, RecommendIterator.java:93-104)."""
    k, n_items, n_users = 32, 150_000, 64
    core, X, Y, rp, col = big_core(k, n_items, n_users, 25, 4711)
    with core:
        before = core.recommend_front_stats()
        bad = []

        def worker(t):
            rng = np.random.default_rng(900 + t)
            for it in range(150):
                if (t + it) % 2 == 0:
                    u = int(rng.integers(0, n_users))
                    ex = col[rp[u]:rp[u + 1]].astype(np.int64)
                    idx, sc, cnt = core.recommend_vectors(X[u:u + 1], 10, exclude=[ex])
                    oidx, osc = to.recommend(Y, X[u], 10, ex)
                else:
                    us = rng.choice(n_users, int(rng.integers(2, 4)), replace=False)
                    idx, sc, cnt = core.recommend_to_many([X[us]], 10)
                    oidx, osc = to.recommend(Y, X[us], 10)
                if not (cnt[0] == len(oidx) and np.array_equal(idx[0, :len(oidx)], oidx)
                        and np.array_equal(sc[0, :len(oidx)].view(np.uint32), np.asarray(osc, np.float32).view(np.uint32))):
                    bad.append((t, it))
        run_threads(16, worker)
        assert not bad, bad[:5]
        st = core.recommend_front_stats()
        assert st["calls"] - before["calls"] == 16 * 150 and st["exclusive"] == before["exclusive"]   # none of them ran alone
        assert st["passes"] - before["passes"] <= 16 * 150
