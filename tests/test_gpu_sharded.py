"""ShardedALS with the real HIP core: two ranks (processes) sharing cuda:0 over gloo must reproduce
the single-rank factors.  (RCCL refuses two ranks on one device, so the collective here is gloo;
the slicing, row offsets, partial Gramians + all-reduce and in-place all-gather are the product
code paths.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import sharded, synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, shape, q, chunk_rows=0):
    n_users, n_items, nnz, k = shape
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, nnz, k, seed=31)
        core = pkg.ALSCore(k, device=0, chunk_rows=chunk_rows)
        core.set_stream(torch.cuda.current_stream().cuda_stream)
        s = sharded.ShardedALS(core, n_users, n_items, k, rank=rank, world=world, device="cuda:0")
        s.set_matrix_from_full(pkg.SIDE_X, *r_csr)
        s.set_matrix_from_full(pkg.SIDE_Y, *c_csr)
        s.set_factors(pkg.SIDE_Y, Y0)
        s.iterate(2)
        torch.cuda.synchronize()
        q.put((rank, s.factors(pkg.SIDE_X).cpu().numpy(), s.factors(pkg.SIDE_Y).cpu().numpy()))
        if world > 1:
            dist.barrier()
        core.close()
    finally:
        if world > 1:
            dist.destroy_process_group()


# chunk_rows 130: 501 users per rank -> 4 pipelined chunks; k = 48: the split-precision gather, whose
# operand scale every rank takes from the largest |value| of the whole matrix (sync_value_bound)
@pytest.mark.parametrize("chunk_rows,k", [(0, 24), (130, 24), (0, 48), (130, 48)])
def test_two_ranks_on_one_gpu_match_single_rank(chunk_rows, k):
    shape = (1001, 333, 30000, k)           # odd sizes: the last slices are padded
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_run, args=(0, 1, 0, shape, q))
    p.start()
    _, X1, Y1 = q.get(timeout=300)
    p.join(timeout=60)
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, shape, q, chunk_rows)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, X, Y in res:
        # partial-Gramian summation order differs from the single-rank Gramian only in fp64 rounding
        assert np.allclose(X, X1, rtol=2e-5, atol=2e-6), rank
        assert np.allclose(Y, Y1, rtol=2e-5, atol=2e-6), rank


def test_chunked_solve_is_bitwise_identical_to_whole_side():
    k, n_users, n_items = 40, 1500, 600
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 40000, k, seed=12)
    outs = []
    for chunk_rows in (0, 400):
        with pkg.ALSCore(k, chunk_rows=chunk_rows, segment_nnz=64) as core:
            core.set_factor_rows(pkg.SIDE_X, n_users)
            core.set_factor_rows(pkg.SIDE_Y, n_items)
            core.set_matrix(pkg.SIDE_X, *r_csr)
            core.set_factors(pkg.SIDE_Y, Y0)
            core.gramian(pkg.SIDE_Y)
            if chunk_rows:
                assert core.num_chunks(pkg.SIDE_X) == 4
                for c in (2, 0, 3, 1):                      # any order
                    core.solve_chunk(pkg.SIDE_X, c)
            else:
                assert core.num_chunks(pkg.SIDE_X) == 1
                core.solve_side(pkg.SIDE_X)
            core.check()
            outs.append(core.get_factors(pkg.SIDE_X))
    assert np.array_equal(outs[0], outs[1])


def _run_rccl_single(port, q, chunk_rows):
    """One rank, backend nccl (= RCCL): world size 1 makes every collective a no-op on the wire, but
    the whole call sequence of the multi-GPU path -- communicator creation on the device, the fp64
    k x k all-reduce, the scalar MAX all-reduce, the in-place all-gather and the chunked async
    all-gathers into strided views, stream ordering against the solve kernels -- runs for real."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n_users, n_items, nnz, k = 1001, 333, 30000, 48
        r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, nnz, k, seed=31)
        out = []
        for force in (False, True):
            core = pkg.ALSCore(k, device=0, chunk_rows=chunk_rows)
            core.set_stream(torch.cuda.current_stream().cuda_stream)
            s = sharded.ShardedALS(core, n_users, n_items, k, rank=0, world=1, device="cuda:0", force_collectives=force)
            s.set_matrix_from_full(pkg.SIDE_X, *r_csr)
            s.set_matrix_from_full(pkg.SIDE_Y, *c_csr)
            s.set_factors(pkg.SIDE_Y, Y0)
            s.iterate(2)
            torch.cuda.synchronize()
            out.append((s.factors(pkg.SIDE_X).cpu().numpy(), s.factors(pkg.SIDE_Y).cpu().numpy()))
            core.close()
        q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("chunk_rows", [0, 260])
def test_rccl_call_sequence_on_one_gpu(chunk_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_run_rccl_single, args=(_free_port(), q, chunk_rows))
    p.start()
    (X0, Y0), (X1, Y1) = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    # partial-Gramian + all-reduce vs the fused Gramian: fp64 summation order only
    assert np.allclose(X1, X0, rtol=2e-5, atol=2e-6) and np.allclose(Y1, Y0, rtol=2e-5, atol=2e-6)
    assert np.all(np.isfinite(X1)) and np.linalg.norm(X1) > 0
