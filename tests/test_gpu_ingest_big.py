"""mals_ingest_finish beyond one sort pipeline (csrc/ingest_big_host.h): user-id range by user-id range, items merged across
ranges, R^T item range by item range -- what lets C5's 5e9 lines (InputFilesReader.readInputFiles, IFR:64-211, is the entry
point of every configuration) through one device.  The partitioned finish must leave EXACTLY what the one-shot pipeline and
the oracle leave: the oracle suites of tests/test_gpu_ingest.py and tests/test_gpu_ingest_text.py run through it with a
partition capacity of a few hundred records (dozens of user ranges, dozens of item ranges), bit for bit."""
import os

import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import _lib, ingest
from tests import test_gpu_ingest as rec
from tests import test_gpu_ingest_text as txt

pytestmark = pytest.mark.gpu
NaN = np.float32("nan")


@pytest.mark.parametrize("n,n_users,n_items,p_remove,part", [
    (65, 2, 2, 0.5, 64), (4097, 50, 40, 0.2, 300), (4097, 50, 40, 0.2, 4096), (20000, 300, 200, 0.1, 700), (50000, 40, 30, 0.05, 4000),
    (100000, 20000, 9000, 0.02, 1500), (100000, 20000, 9000, 0.02, 40000), (300000, 5000, 60000, 0.05, 5000)])
def test_random_streams_in_ranges_match_oracle(n, n_users, n_items, p_remove, part):
    rng = np.random.default_rng(n + n_users + part)
    rec.check(*rec.random_stream(rng, n, n_users, n_items, p_remove), part=part)


def test_heavy_duplicates_wide_ids_and_cancellation_in_ranges():
    rng = np.random.default_rng(5)
    u, i, v = rec.random_stream(rng, 60000, 30, 20, 0.01, values=[1.0, -1.0, 0.5, 2.0, 0.00003, 1e-5])
    rec.check(u, i, v, part=4000)       # 2000 records per user: ranges of one or two users
    n = 30000
    u = rng.integers(-2**62, 2**62, 400).astype(np.int64)[rng.integers(0, 400, n)]
    i = rng.integers(-2**40, 2**40, 300).astype(np.int64)[rng.integers(0, 300, n)]
    v = rng.standard_normal(n).astype(np.float32)
    v[rng.random(n) < 0.1] = NaN
    rec.check(u, i, v, part=1000)


def test_everything_removed_in_ranges():
    n = 2000
    u = np.repeat(np.arange(100, dtype=np.int64), n // 100)
    i = np.tile(np.arange(n // 100, dtype=np.int64), 100)
    v = np.ones(n, np.float32)
    rec.check(np.concatenate([u, u]), np.concatenate([i, i]), np.concatenate([v, np.full(n, NaN, np.float32)]), part=500)


def test_a_user_that_does_not_fit_a_range_is_refused():
    u = np.zeros(5000, np.int64)
    i = np.arange(5000, dtype=np.int64)
    with ingest.Ingest(0) as g:
        g.set_option(_lib.INGEST_OPT_PARTITION_RECORDS, 1000)
        g.append(u, i, np.ones(5000, np.float32))
        with pytest.raises(pkg.MalsError, match="one user id owns too many lines"):
            g.finish()


_MORE = [(3000 + i, 2000 + 41 * (i % 71), [0.05, 0.25, 0.5, 0.75, 0.95][i % 5]) for i in range(int(os.environ.get("MALS_TEXT_SEEDS", "0")) // 4)]


@pytest.mark.parametrize("seed,n_lines,p_odd", [(2, 63, 0.5), (4, 257, 0.6), (5, 5000, 0.25), (6, 40000, 0.1), (7, 3000, 0.9), (8, 20000, 0.5)] + _MORE)
def test_fuzzed_text_corpus_in_ranges_matches_oracle(seed, n_lines, p_odd):
    """tags, removes, knownItemIDs (entries removeSmall pruned included) through the partitioned finish"""
    txt.check([txt.build_corpus(seed, n_lines, p_odd)], part=max(64, n_lines // 9))


def test_ranges_equal_one_pipeline_on_a_million_records():
    """1M records through 23+ user ranges and as many item ranges against the SAME records through one pipeline."""
    rng = np.random.default_rng(77)
    n = 1_000_000
    u = (200_000 * rng.random(n) ** 2).astype(np.int64) * 11 + 5        # active users: the busiest owns ~2 200 records
    i = (50_000 * rng.random(n) ** 3).astype(np.int64) * 3              # popular items: the most popular ~27 000 entries
    v = rng.choice([1.0, 2.0, 0.5, -1.0, 0.00002], n).astype(np.float32)
    v[rng.random(n) < 0.03] = NaN
    res = []
    for part in (None, 60_000):
        with ingest.Ingest(0) as g:
            g.set_option(_lib.INGEST_OPT_KNOWN_ITEMS, 1)
            if part:
                g.set_option(_lib.INGEST_OPT_PARTITION_RECORDS, part)
            g.append(u, i, v)
            g.finish()
            if part:
                assert g.partitions()[0] >= 17 and g.partitions()[1] >= 10, g.partitions()
            res.append((g.counts(), g.ids(pkg.SIDE_X), g.ids(pkg.SIDE_Y), g.csr(pkg.SIDE_X), g.csr(pkg.SIDE_Y), g.known_items()))
    a, b = res
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for x, y in zip(a[3] + a[4] + a[5], b[3] + b[4] + b[5]):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)


def test_partitioned_ingest_feeds_group_and_recommender():
    """the partitioned finish behind mals_ingest_install_group: two iterations on a 2-member group equal the oracle's"""
    from oracle import ingest_text_oracle as to
    from oracle import oracle
    from tests.test_gpu_ingest_group import corpus, rel
    k = 32
    data = corpus(5, 800, 300, 40000)
    want = to.expected([data])
    (uid, rp, col, val), (iid, cp, ccol, cval) = want["csr_x"], want["csr_y"]
    Y0 = (np.random.default_rng(k).standard_normal((len(iid), k)) / np.sqrt(k)).astype(np.float32)
    with ingest.Ingest(0) as g, pkg.GroupALS.single_process(k, [0, 0], backend=_lib.GROUP_PEER_COPY) as grp:
        g.set_option(_lib.INGEST_OPT_PARTITION_RECORDS, 5000)
        g.set_option(_lib.INGEST_OPT_KNOWN_ITEMS, 1)
        g.append_text(data, True)
        g.finish()
        assert g.partitions()[0] >= 8
        g.install_group(grp, copy=True)
        g.close()
        grp.set_factors(pkg.SIDE_Y, Y0)
        grp.iterate(2)
        X = grp.get_factors(pkg.SIDE_X, 0, len(uid))
        Y = grp.get_factors(pkg.SIDE_Y, 0, len(iid))
    Xo, Yo = None, Y0
    for _ in range(2):
        Xo = oracle.half_iteration(rp, col, val, Yo, threads=4)
        Yo = oracle.half_iteration(cp, ccol, cval, Xo, threads=4)
    assert rel(X, Xo) < 1e-4 and rel(Y, Yo) < 1e-4


def _dev_view(torch, ptr, n, dtype, typestr, dev):
    class _V:
        __cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
    return torch.as_tensor(_V(), device=dev) if n else torch.empty(0, dtype=dtype, device=dev)


def _device_csr(torch, g, side, n_rows, nnz, dev):
    import ctypes
    rp, ci, va = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    g._chk(g._L.mals_ingest_device_csr(g._g, side, ctypes.byref(rp), ctypes.byref(ci), ctypes.byref(va)))
    return (_dev_view(torch, rp.value, n_rows + 1, torch.int64, "<i8", dev), _dev_view(torch, ci.value, nnz, torch.int32, "<i4", dev),
            _dev_view(torch, va.value, nnz, torch.float32, "<f4", dev))


def test_two_and_a_half_billion_records_keep_their_properties():
    """2.5e9 records -- more than 2^31: every position that counts across ranges is 64-bit -- through the partitioned finish on
    one device, checked ON the device through size-independent properties (the oracle cannot replay this many): every user
    and item that occurs has a row, the strength of every user and of every item is conserved (integer values: the fp32 sums
    are exact), columns ascend strictly inside every row of both matrices, both matrices hold the same (user, item, value)
    triples, the offsets pass 2^31."""
    import torch
    dev = torch.device("cuda", 0)
    free, total = torch.cuda.mem_get_info()
    if total < 250e9:
        pytest.skip("needs the 288 GB of an MI355X")
    n, n_users, n_items, chunk = 2_500_000_000, 40_000_000, 4_000_000, 250_000_000
    gen = torch.Generator(device=dev).manual_seed(11)
    # (integer strengths summed as int64: exact, and int64 atomics stay fast on the few very popular items -- torch's float64
    # index_add_ falls back to a compare-and-swap loop that crawls under that contention)
    by_user = torch.zeros(n_users, dtype=torch.int64, device=dev)
    by_item = torch.zeros(n_items, dtype=torch.int64, device=dev)
    with ingest.Ingest(0) as g:
        g.set_option(_lib.INGEST_OPT_RESERVE_RECORDS, n)
        for c0 in range(0, n, chunk):
            # popular items, active users: cubes of uniforms, like synth's popularity law
            u = (n_users * torch.rand(chunk, generator=gen, device=dev, dtype=torch.float64) ** 2).long().clamp_(max=n_users - 1)
            i = (n_items * torch.rand(chunk, generator=gen, device=dev, dtype=torch.float64) ** 3).long().clamp_(max=n_items - 1)
            v = torch.randint(1, 4, (chunk,), generator=gen, device=dev).float()
            by_user.index_add_(0, u, v.long())
            by_item.index_add_(0, i, v.long())
            g.append(u, i, v)
            del u, i, v
        torch.cuda.empty_cache()
        g.finish()
        c, st, parts = g.counts(), g.stats(), g.partitions()
        assert c["records"] == n and parts[0] >= 2 and parts[1] >= 2, (c, parts)
        assert c["users"] == int((by_user > 0).sum()) and c["items"] == int((by_item > 0).sum())
        assert c["nnz"] > 2 ** 31, c          # (the pairs are drawn with repeats: fewer entries than records, still beyond 2^31)
        uid = torch.as_tensor(g.ids(pkg.SIDE_X), device=dev)
        iid = torch.as_tensor(g.ids(pkg.SIDE_Y), device=dev)
        assert bool((uid[1:] > uid[:-1]).all()) and bool((iid[1:] > iid[:-1]).all())
        sums = []
        for side, n_rows, ids, want in ((pkg.SIDE_X, c["users"], uid, by_user), (pkg.SIDE_Y, c["items"], iid, by_item)):
            rp, col, val = _device_csr(torch, g, side, n_rows, c["nnz"], dev)
            assert int(rp[0]) == 0 and int(rp[-1]) == c["nnz"] and bool((rp[1:] >= rp[:-1]).all())
            n_cols = c["items"] if side == pkg.SIDE_X else c["users"]
            h_total = torch.zeros((), dtype=torch.int64, device=dev)
            step = max(1, n_rows // 16)
            for r0 in range(0, n_rows, step):
                r1 = min(n_rows, r0 + step)
                e0, e1 = int(rp[r0]), int(rp[r1])
                lens = rp[r0 + 1:r1 + 1] - rp[r0:r1]
                rows = torch.repeat_interleave(torch.arange(r0, r1, device=dev), lens)
                cc = col[e0:e1].long()
                vv = val[e0:e1]
                assert bool((cc >= 0).all()) and bool((cc < n_cols).all())
                # strictly ascending columns inside a row
                same_row = rows[1:] == rows[:-1]
                assert bool(((cc[1:] > cc[:-1]) | ~same_row).all())
                # strength conserved per row (integer-valued fp32 sums are exact): row sums as differences of a running sum
                assert bool((vv == vv.round()).all())
                run = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(vv.long(), 0)])
                got = run[rp[r0 + 1:r1 + 1] - e0] - run[rp[r0:r1] - e0]
                assert torch.equal(got, want[ids[r0:r1]])
                del run
                # the (user, item, value) triples as one wrapping sum of hashes: the same on both sides
                uu, ii = (rows, cc) if side == pkg.SIDE_X else (cc, rows)
                h = (uu * -7046029254386353131 + ii) * -4417276706812531889 + vv.view(torch.int32).long()
                h_total += (h ^ (h >> 29)).sum()
                del rows, cc, vv, lens, same_row, got, uu, ii, h
            sums.append(int(h_total))
            assert int(val.long().sum()) == int(by_user.sum())
        assert sums[0] == sums[1]
    print("ingest 2.5e9 records: %.0f ms, %.2f G records/s, %d user ranges, %d item ranges" % (st["finish_ms"], n / st["finish_ms"] / 1e6, parts[0], parts[1]))
