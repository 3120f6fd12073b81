"""CPU check of the ingest kernels' logic through its lane-level emulation (tests/ingest_emulation.py):
the radix scatter is a stable sort; the composite two-stage sort + replay reproduce the oracle."""
import numpy as np
import pytest

from oracle import ingest_oracle as io
from tests import ingest_emulation as ie


@pytest.mark.parametrize("n,bits", [(1, 8), (63, 8), (700, 8), (1500, 20), (2049, 64)])
def test_radix_sort_is_a_stable_sort(n, bits):
    rng = np.random.default_rng(n)
    hi = (1 << bits) - 1
    keys = rng.integers(0, hi, n, dtype=np.uint64, endpoint=True)
    keys[rng.random(n) < 0.3] = keys[0]                               # plenty of equal keys
    pay = np.arange(n, dtype=np.uint32)
    k, p = ie.radix_sort(keys.copy(), pay.copy())
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order]) and np.array_equal(p, pay[order])


def test_sort_on_the_upper_digits_only_keeps_the_lower_order():
    """the transposed matrix: entries sorted by (row, col), then stably by the col half of (col<<32|row)"""
    rng = np.random.default_rng(3)
    row = np.sort(rng.integers(0, 50, 900)).astype(np.uint64)
    col = rng.integers(0, 300, 900).astype(np.uint64)
    keys = (col << np.uint64(32)) | row
    k, p = ie.radix_sort(keys.copy(), np.arange(900, dtype=np.uint32), first_digit=4)
    assert np.array_equal(k, np.sort(keys, kind="stable")) or np.array_equal(k, keys[np.lexsort((row, col))])


def test_composite_sort_and_replay_reproduce_the_oracle():
    rng = np.random.default_rng(8)
    n = 1200
    u = rng.integers(-5, 40, n).astype(np.int64) * 1000003
    i = rng.integers(0, 25, n).astype(np.int64)
    v = rng.choice([1.0, -1.0, 0.5, 2.0, 0.00003], n).astype(np.float32)
    v[rng.random(n) < 0.1] = np.nan
    flip = np.uint64(1 << 63)
    # stage 1: by item id, payload = record index; stage 2: by user id through that permutation
    k1, p1 = ie.radix_sort(i.astype(np.uint64) ^ flip, np.arange(n, dtype=np.uint32))
    item_rank = np.cumsum(np.concatenate([[1], k1[1:] != k1[:-1]])) - 1
    k2, p2 = ie.radix_sort(u[p1].astype(np.uint64) ^ flip, np.arange(n, dtype=np.uint32))
    idx, ri = p1[p2], item_rank[p2]
    user_rank = np.cumsum(np.concatenate([[1], k2[1:] != k2[:-1]])) - 1
    pair = (user_rank.astype(np.int64) << 32) | ri
    by_row, _ = io.read_input_records(u, i, v)
    heads = np.flatnonzero(np.concatenate([[True], pair[1:] != pair[:-1]]))
    got = {}
    for h, e in zip(heads, list(heads[1:]) + [n]):
        assert np.all(np.diff(idx[h:e].astype(np.int64)) > 0)          # stream order inside the pair
        alive, keep, val = ie.replay_pair(v[idx[h:e]], 1e-4)
        if keep:
            got[(int(u[idx[h]]), int(i[idx[h]]))] = val
    want = {(a, b): val for a, row in by_row.items() for b, val in row.items()}
    assert got.keys() == want.keys()
    assert all(np.float32(got[k]).tobytes() == np.float32(want[k]).tobytes() for k in got)
