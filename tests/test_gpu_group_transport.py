"""The RCCL code path of csrc/mals_group.cpp EXECUTED with N > 1 ranks on the one GPU of the test box.

RCCL refuses two ranks on one device, so these tests point the library at tests/cpp/libmock_rccl.so
(mals_group_use_transport): a stand-in with RCCL's entry points that moves the bytes with plain copies and turns every
mismatched call (a send without its receive, different counts, a rank missing from an all-reduce) into an error
instead of a hang.  What runs is the product's own call sequence -- ncclCommInitAll / ncclCommInitRank, the
grouped ncclSend + ncclRecv exchange per chunk on the comm streams, the k x k and status all-reduces -- and the
factors that come out are checked against the oracle and against the single-device result.  The transport is
loaded once per process, so every case runs in child processes.  The real N-GPU RCCL run is the driver's
scaling bench (bench.py --gpus N)."""
import json
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "cpp", "libmock_rccl.so")
REL_TOL = 1e-4


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def _problem(k, seed):
    from myrrix_recommender_amd import synth
    return synth.numpy_problem(2600, 720, 32000, k, seed=seed, negatives=0.1)


def _in_process(world, k, chunks, q):
    try:
        import myrrix_recommender_amd as pkg
        from myrrix_recommender_amd import _lib
        pkg.GroupALS.use_transport(MOCK)
        r_csr, c_csr, Y0 = _problem(k, 500 + world)
        n_users, n_items = len(r_csr[0]) - 1, len(c_csr[0]) - 1
        with pkg.GroupALS.single_process(k, [0] * world, backend=_lib.GROUP_RCCL, exchange_chunks=chunks) as g:
            g.set_factor_rows(pkg.SIDE_X, n_users)
            g.set_factor_rows(pkg.SIDE_Y, n_items)
            g.set_matrix(pkg.SIDE_X, *r_csr)
            g.set_matrix(pkg.SIDE_Y, *c_csr)
            g.set_factors(pkg.SIDE_Y, Y0)
            g.iterate(2)
            X = g.get_factors(pkg.SIDE_X, 0, n_users)
            Y = g.get_factors(pkg.SIDE_Y, 0, n_items)
            same = all(np.array_equal(g.local(i)[0].get_factors(pkg.SIDE_X), X) and np.array_equal(g.local(i)[0].get_factors(pkg.SIDE_Y), Y)
                       for i in range(world))
            info = [g.comm_info(i) for i in range(world)]       # read back from the communicator
            same = same and [c["comm_size"] for c in info] == [world] * world and [c["comm_rank"] for c in info] == list(range(world))
        q.put(("ok", X, Y, same))
    except Exception as e:  # noqa: BLE001 -- reported to the parent
        q.put(("error", repr(e)))


def _one_rank(rank, world, k, chunks, uid_q, out_q):
    try:
        import myrrix_recommender_amd as pkg
        pkg.GroupALS.use_transport(MOCK)
        r_csr, c_csr, Y0 = _problem(k, 600 + world)
        n_users, n_items = len(r_csr[0]) - 1, len(c_csr[0]) - 1
        if rank == 0:
            uid = pkg.GroupALS.unique_id()
            for _ in range(world - 1):
                uid_q.put(uid)
        else:
            uid = uid_q.get(timeout=120)
        with pkg.GroupALS.from_unique_id(k, 0, world, rank, uid, exchange_chunks=chunks) as g:
            g.set_factor_rows(pkg.SIDE_X, n_users)
            g.set_factor_rows(pkg.SIDE_Y, n_items)
            g.set_matrix(pkg.SIDE_X, *r_csr)
            g.set_matrix(pkg.SIDE_Y, *c_csr)
            g.set_factors(pkg.SIDE_Y, Y0)
            g.iterate(2)
            # every rank reads its own replica: all of them must hold all rows
            X = g.local(0)[0].get_factors(pkg.SIDE_X)
            Y = g.local(0)[0].get_factors(pkg.SIDE_Y)
            bounds = g.bounds(pkg.SIDE_X).tolist()
        out_q.put((rank, "ok", X, Y, bounds))
    except Exception as e:  # noqa: BLE001
        out_q.put((rank, "error", repr(e)))


def _one_rank_gloo(rank, world, k, chunks, port, out_q):
    """The path the driver's multi-GPU bench takes, minus the GPUs: torch.distributed (gloo here) is initialised first,
    GroupALS.from_torch_distributed has rank 0 make the RCCL unique id and broadcast its 128 bytes, every process then
    creates its rank of the group (mals_group_unique_id -> mals_group_create_rank) -- on device 0, over the stand-in."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import myrrix_recommender_amd as pkg
        pkg.GroupALS.use_transport(MOCK)
        r_csr, c_csr, Y0 = _problem(k, 700 + world)
        n_users, n_items = len(r_csr[0]) - 1, len(c_csr[0]) - 1
        with pkg.GroupALS.from_torch_distributed(k, 0, exchange_chunks=chunks) as g:
            g.set_factor_rows(pkg.SIDE_X, n_users)
            g.set_factor_rows(pkg.SIDE_Y, n_items)
            g.set_matrix(pkg.SIDE_X, *r_csr)
            g.set_matrix(pkg.SIDE_Y, *c_csr)
            g.set_factors(pkg.SIDE_Y, Y0)
            g.iterate(2)
            X = g.local(0)[0].get_factors(pkg.SIDE_X)
            Y = g.local(0)[0].get_factors(pkg.SIDE_Y)
            info = g.comm_info(0)
        dist.barrier()
        dist.destroy_process_group()
        out_q.put((rank, "ok", X, Y, info))
    except Exception as e:  # noqa: BLE001
        out_q.put((rank, "error", repr(e)))


def _one_rank_ingest(rank, world, k, data, uid_q, out_q):
    """One process per rank, every rank ingests the SAME input text itself and installs it into its rank of the group
    (mals_ingest_install_group is collective like mals_group_set_matrix): nothing of the matrices crosses a host."""
    try:
        import myrrix_recommender_amd as pkg
        from myrrix_recommender_amd import _lib, ingest
        pkg.GroupALS.use_transport(MOCK)
        if rank == 0:
            uid = pkg.GroupALS.unique_id()
            for _ in range(world - 1):
                uid_q.put(uid)
        else:
            uid = uid_q.get(timeout=120)
        with ingest.Ingest(0) as g, pkg.GroupALS.from_unique_id(k, 0, world, rank, uid, exchange_chunks=2) as grp:
            g.set_option(_lib.INGEST_OPT_KNOWN_ITEMS, 1)
            g.append_text(data, True)
            g.finish()
            c = g.counts()
            g.install_group(grp, copy=True)
            g.close()
            Y0 = (np.random.default_rng(k).standard_normal((c["items"], k)) / np.sqrt(k)).astype(np.float32)
            grp.set_factors(pkg.SIDE_Y, Y0)
            grp.iterate(2)
            core = grp.local(0)[0]
            X, Y = core.get_factors(pkg.SIDE_X), core.get_factors(pkg.SIDE_Y)
            bounds = grp.bounds(pkg.SIDE_X).tolist()
            mine = np.arange(bounds[rank], min(bounds[rank + 1], bounds[rank] + 10), dtype=np.int64)
            rec = core.recommend(mine, 5) if len(mine) else None      # this rank's users: their known items are here
            tags = core.tag_item_count()
        out_q.put((rank, "ok", X, Y, bounds, mine, rec, tags))
    except Exception as e:  # noqa: BLE001
        out_q.put((rank, "error", repr(e)))


def test_ingest_installs_into_a_group_of_one_rank_per_process():
    from oracle import ingest_text_oracle as to
    from oracle import oracle, topn_oracle
    from tests.test_gpu_ingest_group import corpus
    world, k = 3, 32
    data = corpus(21, 700, 260, 30000)
    ctx = mp.get_context("spawn")
    uid_q, out_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_one_rank_ingest, args=(r, world, k, data, uid_q, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out_q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), [r[:3] for r in res if r[1] != "ok"]
    res.sort(key=lambda r: r[0])
    want = to.expected([data])
    (uid, rp, col, val), (iid, cp, ccol, cval) = want["csr_x"], want["csr_y"]
    Xo, Yo = None, (np.random.default_rng(k).standard_normal((len(iid), k)) / np.sqrt(k)).astype(np.float32)
    for _ in range(2):
        Xo = oracle.half_iteration(rp, col, val, Yo, threads=4)
        Yo = oracle.half_iteration(cp, ccol, cval, Xo, threads=4)
    pos = np.searchsorted(iid, want["user_tag_ids"])
    tag_idx = pos[(pos < len(iid)) & (iid[np.minimum(pos, len(iid) - 1)] == want["user_tag_ids"])]
    for rank, _, X, Y, bounds, mine, rec, tags in res:
        assert bounds == res[0][4] and bounds[-1] == len(uid) and tags == len(tag_idx)
        assert np.array_equal(X, res[0][2]) and np.array_equal(Y, res[0][3])
        assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL, (rank, rel(X, Xo), rel(Y, Yo))
        for q, u in enumerate(mine):
            known = want["known_idx"][want["known_ptr"][u]:want["known_ptr"][u + 1]]
            oidx, osc = topn_oracle.recommend(Y, X[u], 5, known, tag_idx)
            assert np.array_equal(rec[0][q, :len(oidx)], oidx) and np.array_equal(rec[1][q, :len(oidx)].view(np.uint32), np.asarray(osc, np.float32).view(np.uint32))


def _oracle(k, seed):
    from oracle import oracle
    r_csr, c_csr, Y0 = _problem(k, seed)
    X, Y = None, Y0
    for _ in range(2):
        X = oracle.half_iteration(*r_csr, Y, threads=4)
        Y = oracle.half_iteration(*c_csr, X, threads=4)
    return X, Y


@pytest.mark.parametrize("world,k,chunks", [(2, 64, 1), (3, 64, 4), (4, 30, 3)])
def test_rccl_backend_members_of_one_process(world, k, chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_in_process, args=(world, k, chunks, q))
    p.start()
    res = q.get(timeout=600)
    p.join(60)
    assert res[0] == "ok", res
    _, X, Y, same = res
    assert same, "the replicas of the members differ after the exchange"
    Xo, Yo = _oracle(k, 500 + world)
    assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL, (rel(X, Xo), rel(Y, Yo))


@pytest.mark.parametrize("world,k,chunks", [(2, 64, 4), (3, 50, 2)])
def test_rccl_backend_one_rank_per_process(world, k, chunks):
    ctx = mp.get_context("spawn")
    uid_q, out_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_one_rank, args=(r, world, k, chunks, uid_q, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out_q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), [r[:3] for r in res if r[1] != "ok"]
    res.sort(key=lambda r: r[0])
    Xo, Yo = _oracle(k, 600 + world)
    for rank, _, X, Y, bounds in res:
        assert bounds == res[0][4] and bounds[0] == 0 and bounds[-1] == X.shape[0]
        assert np.array_equal(X, res[0][2]) and np.array_equal(Y, res[0][3]), "rank %d holds other factors than rank 0" % rank
        assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL, (rank, rel(X, Xo), rel(Y, Yo))


def test_group_through_torch_distributed_two_processes():
    """Two PROCESSES meet exactly as `bench.py --gpus 2` makes them meet: the unique id crosses over torch.distributed,
    each process owns one rank of the group and solves only its slices; both end with the same, oracle-close factors."""
    world, k, chunks = 2, 64, 4
    import socket
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    procs = [ctx.Process(target=_one_rank_gloo, args=(r, world, k, chunks, port, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out_q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), [r[:3] for r in res if r[1] != "ok"]
    res.sort(key=lambda r: r[0])
    Xo, Yo = _oracle(k, 700 + world)
    for rank, _, X, Y, info in res:
        assert info["comm_size"] == world and info["comm_rank"] == rank, info
        assert np.array_equal(X, res[0][2]) and np.array_equal(Y, res[0][3]), "rank %d holds other factors than rank 0" % rank
        assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL, (rank, rel(X, Xo), rel(Y, Yo))


@pytest.mark.parametrize("world", [2, 3])
def test_bench_script_as_the_driver_launches_it(world):
    """`python bench.py --gpus N` end to end (self-launch through torch.distributed.run, one rank per process, the
    group API, the timing bracket, ONE JSON line from rank 0 as the last line of stdout) on the stand-in transport."""
    env = dict(os.environ, MALS_BENCH_TRANSPORT=MOCK, MALS_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                        "--workload", "small"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert sum(ln.startswith("{") for ln in lines) == 1, lines[-5:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "rows/s" and d["value"] > 0 and d["scaling"] == "strong"
    # the same steps timed once more with the status check of every half-iteration inside the region
    assert d["ms_per_step_without_check"] > 0 and d["value_without_check"] > 0
    assert "INVALID_AS_A_MEASUREMENT" in d          # the line says what it is
    assert "RCCL below the C-ABI" in d["config"]["sharding"]
    xb = d["config"]["slices"]["x_bounds"]
    assert len(xb) == world + 1 and xb[0] == 0 and xb[-1] == d["config"]["users"]
    assert d["all_gather_alone_ms"]["x_ms"] > 0
    # the line proves how many ranks the communicator saw and what every rank held
    rk = d["ranks"]
    assert rk["comm_sizes_read_back"] == [world] and len(rk["per_rank"]) == world
    assert sorted(r["comm_rank"] for r in rk["per_rank"]) == list(range(world))
    assert sum(r["x_rows"] for r in rk["per_rank"]) == d["config"]["users"] and sum(r["y_rows"] for r in rk["per_rank"]) == d["config"]["items"]
    assert sum(r["x_nnz"] for r in rk["per_rank"]) == d["config"]["nnz"] == sum(r["y_nnz"] for r in rk["per_rank"])
    assert all(r["pci_bus_id"] for r in rk["per_rank"]) and 0 < rk["ms_per_step_min"] <= rk["ms_per_step_max"]
    assert 0.0 <= d["reconstruction_error"]["mean"] <= 1.0


def test_bench_line_on_one_gpu_says_what_it_leaves_out():
    """`python bench.py` (N = 1) on the small workload: the line carries the time with the per-half status check inside the
    timed region, the CPU baseline sampled by time (work units per thread, CPU model, threads), the fp32-arithmetic leg
    beside the split-f16 headline, and the operand scale / range flag of the last split-precision gather."""
    env = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MALS_BENCH_TRANSPORT", "MALS_BENCH_ONE_DEVICE"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--workload", "small"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    # the dtype names the arithmetic the path really computes in (k = 64 here: split-f16 operands, f32 accumulate)
    assert d["n_gpus"] == 1 and d["dtype"].startswith("f32 storage; per-row Gramian: f16x2-split") and d["vs_baseline"] is None
    assert d["ms_per_step"] > 0 and d["ms_per_step_without_check"] > 0
    assert 0.33 * d["ms_per_step"] < d["ms_per_step_without_check"] < 2.0 * d["ms_per_step"]
    rf, r32 = d["roofline"], d["roofline_fp32"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and 0 < rf["frac"] < 1.2 and rf["peak"] == 8000.0
    assert r32["ms_per_step"] > 0 and 0 < r32["frac"] < 1.2 and "fp32" in r32["workload"]
    assert r32["nnz"] == d["config"]["nnz"]                      # the same problem, another arithmetic
    r24 = d["roofline_split24"]                                  # ... and the three-term f16 split in between (k = 64)
    assert r24["ms_per_step"] > 0 and 0 < r24["frac"] < 1.2 and "split3_f16" in r24["workload"] and r24["nnz"] == d["config"]["nnz"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "rows/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["cpu_model"]
    assert "work units of 100 rows per thread" in cb["sample"]
    gs = d["gather_scale"]
    assert len(gs) == 4 and gs[2] == 1.0 and gs[0] > 0          # the split-f16 kernels ran, not their fp32 twins
    assert "roofline_unplanted" in d
