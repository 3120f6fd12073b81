"""mals_plan_shards (host only): contiguous row slices balanced by cost, not by row count -- the
reference hands out rows in work units of 100 from one queue (ALS:398-408), which balances itself; a
static split has to look at the row lengths."""
import numpy as np

import myrrix_recommender_amd as pkg


def zipf_row_ptr(n_rows, nnz, seed=0):
    """Row lengths in DESCENDING order (dense ids assigned by popularity: the worst case for equal-row slices)."""
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, n_rows + 1) ** 0.8
    lens = rng.multinomial(nnz, w / w.sum())
    lens = np.sort(lens)[::-1]
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


def test_zipf_ordered_ids_are_balanced_by_entries():
    rp = zipf_row_ptr(200_000, 20_000_000)
    for world in (2, 4, 8):
        b = pkg.plan_shards(rp, world, features=64, row_cost=0.0)
        assert b[0] == 0 and b[-1] == 200_000 and np.all(np.diff(b) > 0)
        per = np.diff(rp[b])
        assert per.max() <= 1.02 * per.mean(), (world, per.max() / per.mean())
        # equal-row slices on the same matrix: the first slice alone would hold most of the entries
        eq = np.diff(rp[np.linspace(0, 200_000, world + 1).astype(np.int64)])
        assert eq.max() > 1.5 * eq.mean()


def test_cost_includes_the_rows():
    rp = zipf_row_ptr(100_000, 5_000_000, seed=1)
    b = pkg.plan_shards(rp, 8, features=128)            # default row cost: 128^2 / 200 = 82 entries per row
    cost = np.diff(rp[b]) + 81.92 * np.diff(b)
    assert cost.max() <= 1.02 * cost.mean()
    b0 = pkg.plan_shards(rp, 8, features=128, row_cost=0.0)
    assert not np.array_equal(b, b0)


def test_degenerate_inputs():
    assert list(pkg.plan_shards(np.zeros(1, dtype=np.int64), 4, 64)) == [0, 0, 0, 0, 0]          # no rows
    rp = np.array([0, 0, 0, 0, 0], dtype=np.int64)                                                 # only empty rows
    b = pkg.plan_shards(rp, 2, 64)
    assert b[0] == 0 and b[-1] == 4 and b[1] == 2
    rp = np.array([0, 1000, 1001, 1002], dtype=np.int64)                                           # one giant row
    b = pkg.plan_shards(rp, 3, 64, row_cost=0.0)
    assert list(b) == sorted(b) and b[-1] == 3
    assert list(pkg.plan_shards(rp, 1, 64)) == [0, 3]
