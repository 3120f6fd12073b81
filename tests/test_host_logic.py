"""Host-side logic of the mirror interface that needs no GPU: sparse-matrix mutation semantics
(MatrixUtilsTest.java:33-60), constructor checks (ALS:137-141), initial-Y construction
(ALS:264-335), convergence-sample selection (RandomUtils.java:202-217), work partitioning."""
import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import factorizer, sharded, synth


def test_add_to_semantics():
    byRow, byCol = {}, {}
    assert byRow.get(0) is None and byRow.get(1) is None and byRow.get(4) is None
    pkg.MatrixUtils.addTo(0, 0, -1.0, byRow, byCol)
    pkg.MatrixUtils.addTo(4, 1, 2.0, byRow, byCol)
    assert byRow[0][0] == -1.0 and byCol[0][0] == -1.0
    assert byRow.get(1) is None
    assert byRow[4][1] == 2.0 and byCol[1][4] == 2.0
    assert byRow[4].get(0) is None            # reads as NaN in FastByIDFloatMap
    pkg.MatrixUtils.addTo(4, 1, 0.5, byRow, byCol)    # duplicates sum (FastByIDFloatMap.increment)
    assert byRow[4][1] == 2.5 and byCol[1][4] == 2.5


def test_remove_semantics():
    byRow, byCol = {}, {}
    pkg.MatrixUtils.addTo(0, 0, -1.0, byRow, byCol)
    pkg.MatrixUtils.addTo(4, 1, 2.0, byRow, byCol)
    pkg.MatrixUtils.remove(0, 0, byRow, byCol)
    assert byRow.get(0) is None
    assert byRow[4][1] == 2.0 and byCol[1][4] == 2.0


def test_constructor_preconditions():
    with pytest.raises(ValueError):
        pkg.AlternatingLeastSquares(None, {}, 2, 0.001, 1)
    with pytest.raises(ValueError):
        pkg.AlternatingLeastSquares({}, {}, 0, 0.001, 1)
    for thr in (0.0, 1.0, -1.0):
        with pytest.raises(ValueError):
            pkg.AlternatingLeastSquares({}, {}, 2, thr, 1)
    assert pkg.MatrixFactorizer.DEFAULT_FEATURES == 30


def test_initial_y_reuse_project_pad_and_new_items():
    rng = factorizer.MersenneTwister(0)
    byCol = {10: {1: 1.0}, 11: {1: 1.0}, 12: {2: 1.0}}
    prev = {10: np.array([3.0, 4.0, 0.0], np.float32), 99: np.array([0.0, 0.0, 2.0], np.float32)}
    als = pkg.AlternatingLeastSquares({}, byCol, 3, 0.001, 1)
    als.setPreviousY(prev)
    Y = als._construct_initial_y(rng)
    assert set(Y) == {10, 99, 11, 12}
    assert np.array_equal(Y[10], prev[10])                 # same k: used as is (ALS:304-308)
    for new in (11, 12):
        assert abs(np.linalg.norm(Y[new]) - 1.0) < 1e-6    # random unit vectors (ALS:318-328)
    als2 = pkg.AlternatingLeastSquares({}, byCol, 2, 0.001, 1)   # fewer features: truncate+normalise
    als2.setPreviousY(prev)
    Y2 = als2._construct_initial_y(rng)
    assert np.allclose(Y2[10], [0.6, 0.8])
    als3 = pkg.AlternatingLeastSquares({}, byCol, 5, 0.001, 1)   # more features: pad+normalise
    als3.setPreviousY(prev)
    Y3 = als3._construct_initial_y(rng)
    assert Y3[10].shape == (5,) and abs(np.linalg.norm(Y3[10]) - 1.0) < 1e-6
    assert np.allclose(Y3[10][:3] / Y3[10][0], prev[10] / prev[10][0])


def test_choose_about_n():
    rng = factorizer.MersenneTwister(1)
    assert factorizer._choose_about_n(100, list(range(50)), rng) == list(range(50))
    picks = [len(factorizer._choose_about_n(100, list(range(100000)), rng)) for _ in range(20)]
    assert 60 < np.mean(picks) < 140
    one = factorizer._choose_about_n(100, list(range(100000)), rng)
    assert one == sorted(set(one)) and one[-1] < 100000


def test_rows_per_rank_and_slices_cover_everything():
    class FakeCore:
        def bind_factors(self, side, t):
            pass
    for n_users, n_items, world in [(10, 7, 2), (1000, 33, 8), (5, 5, 8)]:
        covered_u, covered_i = [], []
        for rank in range(world):
            s = sharded.ShardedALS(FakeCore(), n_users, n_items, 4, rank=rank, world=world, device="cpu")
            covered_u.extend(range(*s.slice_bounds(pkg.SIDE_X)))
            covered_i.extend(range(*s.slice_bounds(pkg.SIDE_Y)))
            assert s.F[pkg.SIDE_X].shape[0] == world * sharded.rows_per_rank(n_users, world)
        assert covered_u == list(range(n_users)) and covered_i == list(range(n_items))


def test_synth_problem_is_consistent():
    r_csr, c_csr, Y0 = synth.numpy_problem(200, 80, 3000, 6, seed=3)
    assert r_csr[0][-1] == c_csr[0][-1] == len(r_csr[1]) == len(c_csr[1])
    R = np.zeros((200, 80), np.float32)
    for u in range(200):
        s, e = r_csr[0][u], r_csr[0][u + 1]
        assert len(set(r_csr[1][s:e])) == e - s          # distinct columns per row
        R[u, r_csr[1][s:e]] = r_csr[2][s:e]
    Rt = np.zeros((80, 200), np.float32)
    for i in range(80):
        s, e = c_csr[0][i], c_csr[0][i + 1]
        Rt[i, c_csr[1][s:e]] = c_csr[2][s:e]
    assert np.array_equal(R.T, Rt)
    assert np.allclose(np.linalg.norm(Y0, axis=1), 1.0, atol=1e-6)
