"""The multi-GPU half-iteration below the C-ABI (mals_group_*, csrc/mals_group.cpp) against the oracle.

The GPU box has ONE device, so the N > 1 orchestration -- cost-balanced slices, per-side chunking, partial
Gramians + k x k sum, chunk-pipelined exchange into every replica, agreed status -- runs here with N members
on device 0 and the peer-copy backend (RCCL refuses two ranks on one device); the RCCL backend runs with
a real one-rank communicator (ncclCommInitRank + ncclAllReduce on it).  The N-GPU RCCL run itself is the
driver's scaling bench."""
import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import _lib, synth
from oracle import oracle

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def oracle_iterations(r_csr, c_csr, Y0, n, **kw):
    X, Y = None, Y0
    for _ in range(n):
        X = oracle.half_iteration(*r_csr, Y, threads=4, **kw)
        Y = oracle.half_iteration(*c_csr, X, threads=4, **kw)
    return X, Y


@pytest.mark.parametrize("world,k,chunks", [(1, 64, 4), (2, 64, 1), (3, 64, 4), (4, 50, 3), (3, 128, 2)])
def test_peer_copy_group_matches_oracle(world, k, chunks):
    n_users, n_items = 2500, 700
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 30000, k, seed=300 + world, negatives=0.1)
    with pkg.GroupALS.single_process(k, [0] * world, backend=_lib.GROUP_PEER_COPY, exchange_chunks=chunks) as g:
        g.set_factor_rows(pkg.SIDE_X, n_users)
        g.set_factor_rows(pkg.SIDE_Y, n_items + 5)       # five stale Y rows (ALS:304-308): they count in Y^T Y
        g.set_matrix(pkg.SIDE_X, *r_csr)
        g.set_matrix(pkg.SIDE_Y, *c_csr)
        stale = np.random.default_rng(1).standard_normal((5, k)).astype(np.float32) * 0.1
        g.set_factors(pkg.SIDE_Y, np.vstack([Y0, stale]))
        g.iterate(2)
        X = g.get_factors(pkg.SIDE_X, 0, n_users)
        Y = g.get_factors(pkg.SIDE_Y, 0, n_items + 5)
        bx, by = g.bounds(pkg.SIDE_X), g.bounds(pkg.SIDE_Y)
        # every replica holds the same factors
        for i in range(world):
            core, rank = g.local(i)
            assert rank == i
            assert np.array_equal(core.get_factors(pkg.SIDE_X), X)
            assert np.array_equal(core.get_factors(pkg.SIDE_Y), Y)
    assert bx[0] == 0 and bx[-1] == n_users and np.all(np.diff(bx) >= 0) and by[-1] == n_items
    # oracle: stale rows never re-solved, always in the Gramian
    Xo, Yo = None, np.vstack([Y0, stale])
    for _ in range(2):
        Xo = oracle.half_iteration(*r_csr, Yo, threads=4)
        Yn = oracle.half_iteration(*c_csr, Xo, threads=4)
        Yo = np.vstack([Yn, stale])
    assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL, (world, rel(X, Xo), rel(Y, Yo))
    assert np.array_equal(Y[n_items:], stale)


@pytest.mark.parametrize("k,world", [(64, 1), (64, 3), (128, 2), (50, 2)])
def test_chunks_on_alternating_streams_leave_the_factors_bitwise_alone(k, world):
    """Consecutive chunks of a half-iteration run on two alternating compute streams (a chunk's tail overlaps the next
    chunk's head; mals_group_set_alternate_streams).  Neither that nor the chunking itself may change a single bit: every
    row is solved from the same inputs by the same kernel whichever chunk and stream it is in.  4 chunks on two streams vs
    4 chunks on one vs 1 chunk, three iterations, every path live (dual rows, direct rows, long rows in fixed 64-entry
    segments, refinement of planted ill-conditioned rows)."""
    n_users, n_items = 6000, 900
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 90000, k, seed=4242 + k, negatives=0.05)
    res = {}
    for name, chunks, alternate in (("two streams", 4, True), ("one stream", 4, False), ("one chunk", 1, True)):
        with pkg.GroupALS.single_process(k, [0] * world, backend=_lib.GROUP_PEER_COPY, exchange_chunks=chunks, segment_nnz=64) as g:
            g.set_alternate_streams(alternate)
            g.set_factor_rows(pkg.SIDE_X, n_users)
            g.set_factor_rows(pkg.SIDE_Y, n_items)
            g.set_matrix(pkg.SIDE_X, *r_csr)
            g.set_matrix(pkg.SIDE_Y, *c_csr)
            g.set_factors(pkg.SIDE_Y, Y0)
            g.iterate(3)
            res[name] = (g.get_factors(pkg.SIDE_X, 0, n_users), g.get_factors(pkg.SIDE_Y, 0, n_items))
            st = g.local(0)[0].stats()
            assert st["rows_solved"] > 0
    for name in ("one stream", "one chunk"):
        assert np.array_equal(res["two streams"][0], res[name][0]), (name, rel(res["two streams"][0], res[name][0]))
        assert np.array_equal(res["two streams"][1], res[name][1]), (name, rel(res["two streams"][1], res[name][1]))
    Xo, Yo = oracle_iterations(r_csr, c_csr, Y0, 3)
    assert rel(res["two streams"][0], Xo) < REL_TOL and rel(res["two streams"][1], Yo) < REL_TOL


def test_group_result_does_not_depend_on_the_sharding():
    k, n_users, n_items = 64, 2000, 600
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 25000, k, seed=77)
    res = []
    for world in (1, 3):
        with pkg.GroupALS.single_process(k, [0] * world, backend=_lib.GROUP_PEER_COPY) as g:
            g.set_factor_rows(pkg.SIDE_X, n_users)
            g.set_factor_rows(pkg.SIDE_Y, n_items)
            g.set_matrix(pkg.SIDE_X, *r_csr)
            g.set_matrix(pkg.SIDE_Y, *c_csr)
            g.set_factors(pkg.SIDE_Y, Y0)
            g.iterate(2)
            res.append((g.get_factors(pkg.SIDE_X, 0, n_users), g.get_factors(pkg.SIDE_Y, 0, n_items)))
    # same operand scales on every rank (value statistics are summed over the group); only the k x k sum of
    # the partial Gramians is ordered differently
    assert rel(res[1][0], res[0][0]) < 2e-6 and rel(res[1][1], res[0][1]) < 2e-6


def test_chunked_upload_and_factorize_match_single_handle_call():
    """The JNI call sequence on a group: begin/append/end with Java-sized pieces, then call()."""
    k, n_users, n_items = 10, 943, 1682     # C1 shape (MovieLens 100K, k = 10)
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 100000, k, seed=5)
    tu = np.arange(0, n_users, 9, dtype=np.int64)[:100]
    ti = np.arange(0, n_items, 17, dtype=np.int64)[:100]
    with pkg.GroupALS.single_process(k, [0, 0], backend=_lib.GROUP_PEER_COPY) as g:
        g.set_factor_rows(pkg.SIDE_X, n_users)
        g.set_factor_rows(pkg.SIDE_Y, n_items)
        g.set_matrix_chunked(pkg.SIDE_X, *r_csr, rows_per_piece=100)
        g.set_matrix_chunked(pkg.SIDE_Y, *c_csr, rows_per_piece=333)
        g.set_factors(pkg.SIDE_Y, Y0)
        iters, conv = g.factorize(0.001, 6, False, True, tu, ti)
        X = g.get_factors(pkg.SIDE_X, 0, n_users)
        Y = g.get_factors(pkg.SIDE_Y, 0, n_items)
    Xo, Yo, it_o, conv_o = oracle.als_call(r_csr, c_csr, n_users, n_items, Y0, k, conv_threshold=0.001, max_iterations=6,
                                           test_users=tu, test_items=ti, threads=4)
    assert iters == it_o
    assert abs(conv - conv_o) <= 1e-4 * max(abs(conv_o), 1e-12)
    assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL


def test_singular_row_is_reported_by_the_group():
    k = 8
    rp = np.array([0, 1, 2, 2], dtype=np.int64)          # the last row is empty: W = G, rank 2 < 8
    col = np.array([0, 1], dtype=np.int32)
    val = np.ones(2, dtype=np.float32)
    with pkg.GroupALS.single_process(k, [0, 0], backend=_lib.GROUP_PEER_COPY, lam=0.0) as g:
        g.set_factor_rows(pkg.SIDE_X, 3)
        g.set_factor_rows(pkg.SIDE_Y, 2)
        g.set_matrix(pkg.SIDE_X, rp, col, val)
        g.set_factors(pkg.SIDE_Y, np.eye(2, k, dtype=np.float32))
        with pytest.raises(pkg.SingularSystem):
            g.half_iteration(pkg.SIDE_X)


def test_rccl_one_rank_communicator():
    """RCCL itself: unique id, ncclCommInitRank, the value-statistics and Gramian all-reduces on a
    one-rank communicator, the agreed status."""
    k, n_users, n_items = 64, 1500, 500
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 20000, k, seed=8)
    with pkg.GroupALS.from_torch_distributed(k, 0, world=1, rank=0, one_rank_communicator=True) as g:
        g.set_factor_rows(pkg.SIDE_X, n_users)
        g.set_factor_rows(pkg.SIDE_Y, n_items)
        g.set_matrix(pkg.SIDE_X, *r_csr)
        g.set_matrix(pkg.SIDE_Y, *c_csr)
        g.set_factors(pkg.SIDE_Y, Y0)
        g.iterate(1)
        X = g.get_factors(pkg.SIDE_X, 0, n_users)
        Y = g.get_factors(pkg.SIDE_Y, 0, n_items)
    Xo, Yo = oracle_iterations(r_csr, c_csr, Y0, 1)
    assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL


def test_device_matrix_and_member_views():
    torch = pytest.importorskip("torch")
    k, n_users, n_items = 64, 1200, 400
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 15000, k, seed=9)
    dev = torch.device("cuda", 0)
    with pkg.GroupALS.single_process(k, [0, 0, 0], backend=_lib.GROUP_PEER_COPY) as g:
        g.set_factor_rows(pkg.SIDE_X, n_users)
        g.set_factor_rows(pkg.SIDE_Y, n_items)
        g.set_matrix(pkg.SIDE_X, *[torch.as_tensor(a).to(dev) for a in r_csr])
        g.set_matrix(pkg.SIDE_Y, *[torch.as_tensor(a).to(dev) for a in c_csr])
        g.set_factors(pkg.SIDE_Y, Y0)
        g.iterate(1)
        X = g.get_factors(pkg.SIDE_X, 0, n_users)
        solved = sum(g.local(i)[0].stats()["rows_solved"] for i in range(3))
    assert solved == n_users + n_items
    Xo, _ = oracle_iterations(r_csr, c_csr, Y0, 1)
    assert rel(X, Xo) < REL_TOL


@pytest.mark.parametrize("k", [64, 128])
def test_members_are_not_serialised_behind_host_work(k):
    """One process driving N GPUs (the JVM's deployment, jni/myrrix_als_jni.c -> mals_group_create) is the reference's
    one pool whose workers all start before any result is awaited (ALS:186-191,391-410): every member's direct kernels
    are enqueued before ANY member's host work -- the k x k eigendecomposition of the dual path, 4 ms at k = 128 --
    starts, and that decomposition is computed ONCE per half-iteration (the all-reduced G is the same everywhere),
    not once per member."""
    world, n_users, n_items = 4, 4000, 1200
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 60000, k, seed=1234, negatives=0.05)
    with pkg.GroupALS.single_process(k, [0] * world, backend=_lib.GROUP_PEER_COPY, exchange_chunks=2) as g:
        g.set_factor_rows(pkg.SIDE_X, n_users)
        g.set_factor_rows(pkg.SIDE_Y, n_items)
        g.set_matrix(pkg.SIDE_X, *r_csr)
        g.set_matrix(pkg.SIDE_Y, *c_csr)
        g.set_factors(pkg.SIDE_Y, Y0)
        cores = [g.local(i)[0] for i in range(world)]
        for side in (pkg.SIDE_X, pkg.SIDE_Y):
            for c in cores:
                c.reset_stats()
            g.half_iteration(side)
            tl = [c.timeline() for c in cores]
            st = [c.stats() for c in cores]
            assert all(s["rows_dual"] > 0 for s in st), "the short rows of every member go through the dual kernels"
            computed = [i for i in range(world) if st[i]["eigen_host_ms"] > 0.0]
            assert computed == [0], "one eigendecomposition per group half-iteration, on member 0: %r" % (computed,)
            first_host_work = min(t[1] for t in tl)
            assert first_host_work > 0.0
            for i, t in enumerate(tl):
                assert 0.0 < t[0] <= first_host_work, "member %d's first kernel was enqueued %.0f us AFTER host work began" % (i, t[0] - first_host_work)
                assert t[1] <= t[2]
        X = g.get_factors(pkg.SIDE_X, 0, n_users)
        Y = g.get_factors(pkg.SIDE_Y, 0, n_items)
    Xo = oracle.half_iteration(*r_csr, Y0, threads=4)
    Yo = oracle.half_iteration(*c_csr, Xo, threads=4)
    assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL


def test_exchange_chunks_changed_after_the_upload_cannot_leave_rows_unsolved():
    """mals_group_set_exchange_chunks after a matrix is set used to shorten the solve loop under the members' work
    lists: the count now belongs to the upload."""
    k, n_users, n_items = 64, 3000, 800
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 40000, k, seed=4321)
    with pkg.GroupALS.single_process(k, [0, 0], backend=_lib.GROUP_PEER_COPY, exchange_chunks=4) as g:
        g.set_factor_rows(pkg.SIDE_X, n_users)
        g.set_factor_rows(pkg.SIDE_Y, n_items)
        g.set_matrix(pkg.SIDE_X, *r_csr)
        g.set_matrix(pkg.SIDE_Y, *c_csr)
        g.set_factors(pkg.SIDE_Y, Y0)
        g._chk(g._L.mals_group_set_exchange_chunks(g._g, 2))      # lower it under the uploaded X side ...
        g.half_iteration(pkg.SIDE_X)
        X = g.get_factors(pkg.SIDE_X, 0, n_users)
        g.set_matrix(pkg.SIDE_Y, *c_csr)                          # ... and let the Y side be uploaded with the new count
        g.half_iteration(pkg.SIDE_Y)
        Y = g.get_factors(pkg.SIDE_Y, 0, n_items)
        assert sum(g.local(i)[0].stats()["rows_solved"] for i in range(2)) == n_users + n_items
        assert g.local(0)[0].num_chunks(pkg.SIDE_X) == 4 and g.local(0)[0].num_chunks(pkg.SIDE_Y) == 2
    Xo = oracle.half_iteration(*r_csr, Y0, threads=4)
    Yo = oracle.half_iteration(*c_csr, Xo, threads=4)
    assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL
