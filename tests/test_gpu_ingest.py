"""the device ingest -> CSR path (mals_ingest_*) against the oracle's
record-by-record restatement of InputFilesReader / MatrixUtils.  Integer and index outputs must be
identical; values are bit-exact too (one fp32 add per record, in file order, on both sides)."""
import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import _lib, ingest
from oracle import ingest_oracle as io

pytestmark = pytest.mark.gpu

NaN = np.float32("nan")


def check(u, i, v, thr=1.0e-4, part=None):
    """part: MALS_INGEST_OPT_PARTITION_RECORDS -- the finish then runs user-id range by user-id range (ingest_big_host.h)"""
    (uid, rp, col, val), (iid, cp, ccol, cval) = io.expected_matrices(u, i, v, thr)
    with ingest.Ingest(0, thr) as g:
        if part:
            g.set_option(_lib.INGEST_OPT_PARTITION_RECORDS, part)
        g.append(u, i, v)
        g.finish()
        if part and len(u) > part:
            assert g.partitions()[0] >= 2, g.partitions()
        c = g.counts()
        assert c == {"records": len(u), "users": len(uid), "items": len(iid), "nnz": len(col)}
        assert np.array_equal(g.ids(pkg.SIDE_X), uid) and np.array_equal(g.ids(pkg.SIDE_Y), iid)
        grp, gcol, gval = g.csr(pkg.SIDE_X)
        assert np.array_equal(grp, rp) and np.array_equal(gcol, col)
        assert np.array_equal(gval.view(np.uint32), val.view(np.uint32))
        gcp, gccol, gcval = g.csr(pkg.SIDE_Y)
        assert np.array_equal(gcp, cp) and np.array_equal(gccol, ccol)
        assert np.array_equal(gcval.view(np.uint32), cval.view(np.uint32))
        return g.stats()


def random_stream(rng, n, n_users, n_items, p_remove=0.1, id_scale=1, values=None):
    u = rng.integers(0, n_users, n).astype(np.int64) * id_scale
    i = rng.integers(0, n_items, n).astype(np.int64) * id_scale
    v = (rng.choice(values, n) if values is not None else rng.standard_normal(n) * 3).astype(np.float32)
    v[rng.random(n) < p_remove] = NaN
    return u, i, v


def test_reference_unit_cases():
    check(np.array([0, 4], np.int64), np.array([0, 1], np.int64), np.array([-1.0, 2.0], np.float32))        # testAddTo
    check(np.array([0, 4, 0], np.int64), np.array([0, 1, 0], np.int64), np.array([-1.0, 2.0, NaN], np.float32))  # testRemove
    check(np.array([5, 5, 6, 5], np.int64), np.array([9, 9, 9, 8], np.int64), np.array([1.0, -1.0, 2.0, 0.00005], np.float32))


@pytest.mark.parametrize("n,n_users,n_items,p_remove", [
    (1, 1, 1, 0.0), (63, 5, 4, 0.3), (64, 7, 3, 0.0), (65, 2, 2, 0.5), (4097, 50, 40, 0.2),
    (20000, 300, 200, 0.1), (50000, 40, 30, 0.05), (100000, 20000, 9000, 0.02)])
def test_random_streams_match_oracle(n, n_users, n_items, p_remove):
    rng = np.random.default_rng(n + n_users)
    check(*random_stream(rng, n, n_users, n_items, p_remove))


def test_heavy_duplicates_and_cancellation():
    """Many records per pair, values that cancel below the threshold, integer-like strengths."""
    rng = np.random.default_rng(5)
    u, i, v = random_stream(rng, 60000, 30, 20, 0.01, values=[1.0, -1.0, 0.5, 2.0, 0.00003, 1e-5])
    check(u, i, v)


def test_wide_and_negative_ids():
    """64-bit ids: hashed tags (OneWayMigrator) are arbitrary longs, negative included."""
    rng = np.random.default_rng(9)
    n = 30000
    u = rng.integers(-2**62, 2**62, 400).astype(np.int64)[rng.integers(0, 400, n)]
    i = rng.integers(-2**40, 2**40, 300).astype(np.int64)[rng.integers(0, 300, n)]
    v = rng.standard_normal(n).astype(np.float32)
    v[rng.random(n) < 0.1] = NaN
    st = check(u, i, v)
    assert st["radix_passes"] >= 8                   # the high digits are not trivial here


def test_everything_removed_and_empty_input():
    u = np.array([1, 2, 1, 2], np.int64)
    i = np.array([1, 1, 1, 1], np.int64)
    v = np.array([1.0, 2.0, NaN, NaN], np.float32)
    check(u, i, v)
    with ingest.Ingest(0) as g:
        g.finish()
        assert g.counts() == {"records": 0, "users": 0, "items": 0, "nnz": 0}
        assert g.csr(pkg.SIDE_X)[0].tolist() == [0]


def test_appending_in_batches_equals_one_batch():
    rng = np.random.default_rng(12)
    u, i, v = random_stream(rng, 30000, 500, 400, 0.1)
    with ingest.Ingest(0) as a, ingest.Ingest(0) as b:
        a.append(u, i, v)
        for lo in range(0, len(u), 7001):
            b.append(u[lo:lo + 7001], i[lo:lo + 7001], v[lo:lo + 7001])
        a.finish()
        b.finish()
        for side in (pkg.SIDE_X, pkg.SIDE_Y):
            assert all(np.array_equal(x, y) for x, y in zip(a.csr(side), b.csr(side)))


def test_ingest_feeds_the_factorizer():
    """readInputFiles -> runFactorization (DGM:333-355): the matrices stay on the device."""
    rng = np.random.default_rng(3)
    n_users, n_items, k = 400, 150, 24
    u = rng.integers(0, n_users, 20000).astype(np.int64) + 1000
    i = rng.integers(0, n_items, 20000).astype(np.int64) + 50
    v = rng.choice([1.0, 2.0, 3.0], 20000).astype(np.float32)
    (uid, rp, col, val), (iid, cp, ccol, cval) = io.expected_matrices(u, i, v)
    from oracle import oracle
    Y0 = (rng.standard_normal((len(iid), k)) / np.sqrt(k)).astype(np.float32)
    with ingest.Ingest(0) as g, pkg.ALSCore(k) as core:
        g.append(u, i, v)
        g.finish()
        core.set_factor_rows(pkg.SIDE_X, len(uid))
        core.set_factor_rows(pkg.SIDE_Y, len(iid))
        g.install(core)
        core.set_factors(pkg.SIDE_Y, Y0)
        core.half_iteration(pkg.SIDE_X)
        core.half_iteration(pkg.SIDE_Y)
        X, Y = core.get_factors(pkg.SIDE_X), core.get_factors(pkg.SIDE_Y)
    Xo = oracle.half_iteration(rp, col, val, Y0)
    Yo = oracle.half_iteration(cp, ccol, cval, Xo)
    assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < 1e-4
    assert np.linalg.norm(Y - Yo) / np.linalg.norm(Yo) < 1e-4


def test_large_stream_properties():
    """2e8 records (size-independent properties; the oracle cannot replay this many)."""
    import torch
    n, n_users, n_items = 200_000_000, 3_000_000, 400_000
    gen = torch.Generator(device="cuda").manual_seed(7)
    u = torch.randint(0, n_users, (n,), device="cuda", generator=gen)
    i = torch.randint(0, n_items, (n,), device="cuda", generator=gen)
    v = torch.randint(1, 6, (n,), device="cuda", generator=gen).float()
    with ingest.Ingest(0) as g:
        g.append(u, i, v)
        g.finish()
        c, st = g.counts(), g.stats()
        rp, col, val = g.csr(pkg.SIDE_X)
        cp, ccol, cval = g.csr(pkg.SIDE_Y)
    assert c["users"] == len(torch.unique(u)) and c["items"] == len(torch.unique(i))
    key = u * n_items + i
    assert c["nnz"] == len(torch.unique(key))                       # no removes, integer values: every pair survives
    assert rp[0] == 0 and rp[-1] == c["nnz"] and np.all(np.diff(rp) >= 0) and cp[-1] == c["nnz"]
    # columns strictly ascending within every row (sorted, duplicates merged)
    d = np.diff(col.astype(np.int64))
    row_start = np.zeros(len(col), bool)
    row_start[rp[1:-1][rp[1:-1] < len(col)]] = True
    assert np.all((d > 0) | row_start[1:])
    # the total strength is conserved, by row and by column (integer-valued: exact in fp32 sums here)
    assert float(val.astype(np.float64).sum()) == float(v.double().sum().item()) == float(cval.astype(np.float64).sum())
    # transposition consistency: same multiset of (row, col, value)
    rows = np.repeat(np.arange(c["users"], dtype=np.int64), np.diff(rp))
    crow = np.repeat(np.arange(c["items"], dtype=np.int64), np.diff(cp))
    a = np.lexsort((rows, col))
    assert np.array_equal(col[a], crow) and np.array_equal(rows[a], ccol) and np.array_equal(val[a], cval)
    print("ingest 2e8 records: %.1f ms, %.0f M records/s, %d radix passes, %.0f GB/s algorithmic" %
          (st["finish_ms"], n / st["finish_ms"] / 1e3, st["radix_passes"], st["bytes_moved"] / st["finish_ms"] / 1e6))


@pytest.mark.parametrize("seed", range(60))
def test_seeded_stream_sweep(seed):
    """Random record streams over the whole shape space (sizes around the sort's tile boundaries, few or many ids,
    wide and negative ids, removal and cancellation rates, value sets that cancel exactly, thresholds) -- bit-exact
    against the oracle's replay of MatrixUtils.addTo / remove."""
    rng = np.random.default_rng(31_000 + seed)
    n = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1023, 4096, 4097, 10000, 65536, 65537, 150000]))
    n_users = int(rng.choice([1, 2, 7, 100, 5000, 100000]))
    n_items = int(rng.choice([1, 3, 50, 3000, 80000]))
    id_scale = int(rng.choice([1, 1, 7919, -3, 2**40 + 11]))
    values = [None, None, [1.0, -1.0, 0.5, 2.0, 0.00003, 1e-5], [1.0, -1.0], [0.0, 1e-7, -1e-7, 3.0]][int(rng.integers(0, 5))]
    u, i, v = random_stream(rng, n, n_users, n_items, float(rng.choice([0.0, 0.02, 0.3, 0.9])), id_scale, values)
    if rng.random() < 0.2:   # sorted input, as a log replay would deliver it
        order = np.lexsort((i, u))
        u, i, v = u[order], i[order], v[order]
    check(u, i, v, thr=float(rng.choice([1.0e-4, 1.0e-4, 0.0, 0.5])))
