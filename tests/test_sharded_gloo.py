"""N>1 path on CPU: world_size-2 gloo run of ShardedALS.  The per-rank compute is a stand-in backed
by the oracle (this is a test of the sharding / exchange logic, which is all ShardedALS adds over
ALSCore); the result must equal the unsharded oracle iteration."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import sharded, synth
from oracle import oracle


class OracleBackedCore:
    """Duck-types the ALSCore methods ShardedALS uses, on CPU tensors."""

    def __init__(self, k, alpha=1.0, lam=0.1, chunk_rows=0):
        self.k, self.alpha, self.lam = k, alpha, lam
        self.chunk_rows = chunk_rows
        self.F, self.M, self.G, self.bound = {}, {}, {}, {}

    def bind_factors(self, side, t):
        self.F[side] = t

    def set_matrix(self, side, row_ptr, col, val, row_offset=0):
        self.M[side] = (np.asarray(row_ptr), np.asarray(col), np.asarray(val), row_offset)

    def value_bound(self, side):
        return self.bound.get(side, float(np.max(np.abs(self.M[side][2]), initial=0.0)))

    def set_value_bound(self, side, v):
        assert v >= float(np.max(np.abs(self.M[side][2]), initial=0.0))
        self.bound[side] = v

    def gramian(self, side):
        self.G[side] = oracle.gramian(self.F[side].numpy())

    def gramian_partial(self, side, row_begin, n_rows, out):
        out.copy_(torch.from_numpy(oracle.gramian(self.F[side][row_begin:row_begin + n_rows].numpy())))

    def set_gramian(self, side, G):
        self.G[side] = G.numpy().copy()

    def solve_side(self, side):
        rp, col, val, off = self.M[side]
        out = oracle.solve_rows(rp, col, val, self.F[1 - side].numpy(), self.G[1 - side],
                                alpha=self.alpha, lam=self.lam)
        self.F[side][off:off + len(rp) - 1].copy_(torch.from_numpy(out))

    def num_chunks(self, side):
        n = len(self.M[side][0]) - 1
        return max(1, -(-n // self.chunk_rows))

    def solve_chunk(self, side, c):
        rp, col, val, off = self.M[side]
        n = len(rp) - 1
        r0, r1 = min(n, c * self.chunk_rows), min(n, (c + 1) * self.chunk_rows)
        out = oracle.solve_rows(rp, col, val, self.F[1 - side].numpy(), self.G[1 - side],
                                alpha=self.alpha, lam=self.lam, row_begin=r0, row_end=r1)
        self.F[side][off + r0:off + r1].copy_(torch.from_numpy(out[r0:r1]))

    def check(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_users, n_items, k, nnz, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, nnz, k, seed=9)
        chunk_rows = 13 if mode == "chunked" else 0        # 51 users per rank -> 4 chunks; 19 items -> 2
        s = sharded.ShardedALS(OracleBackedCore(k, chunk_rows=chunk_rows), n_users, n_items, k, rank=rank, world=world,
                               device="cpu", gramian_mode="allreduce" if mode == "chunked" else mode)
        s.set_matrix_from_full(pkg.SIDE_X, *r_csr)
        s.set_matrix_from_full(pkg.SIDE_Y, *c_csr)
        s.set_factors(pkg.SIDE_Y, Y0)
        s.iterate(2)
        # one operand-scale bound for all ranks: the largest |value| of the WHOLE matrix
        assert s.core.bound[pkg.SIDE_X] == s.core.bound[pkg.SIDE_Y] == float(np.max(np.abs(r_csr[2])))
        q.put((rank, s.factors(pkg.SIDE_X).numpy().copy(), s.factors(pkg.SIDE_Y).numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "replicated", "chunked"])
def test_world2_gloo_matches_unsharded(mode):
    n_users, n_items, k, nnz, world = 101, 37, 6, 1500, 2     # odd sizes: last slice is padded
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_users, n_items, k, nnz, mode, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, nnz, k, seed=9)
    Xo, Yo = None, Y0
    for _ in range(2):
        Xo = oracle.half_iteration(*r_csr, Yo)
        Yo = oracle.half_iteration(*c_csr, Xo)
    for rank, X, Y in results:
        # every rank ends with the same full replicas; partial-Gramian summation order only moves
        # fp64 rounding, far below fp32 storage
        assert np.allclose(X, Xo, rtol=1e-5, atol=1e-6), rank
        assert np.allclose(Y, Yo, rtol=1e-5, atol=1e-6), rank


def _worker_singular(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_users, n_items, k = 40, 20, 4
        r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 300, k, seed=3)

        class FailsOnRank1(OracleBackedCore):
            def check(self):
                if rank == 1:     # the rank that owns the singular row (CMLSS:46-54 on that rank only)
                    raise pkg.SingularSystem(1, "near-singular system for row 33 of side X", 0, 33, 2)

        s = sharded.ShardedALS(FailsOnRank1(k), n_users, n_items, k, rank=rank, world=world, device="cpu")
        s.set_matrix_from_full(pkg.SIDE_X, *r_csr)
        s.set_matrix_from_full(pkg.SIDE_Y, *c_csr)
        s.set_factors(pkg.SIDE_Y, Y0)
        try:
            s.iterate(1)
            q.put((rank, "no error"))
        except pkg.SingularSystem as e:
            q.put((rank, "singular row %d" % e.row))
        except RuntimeError as e:
            q.put((rank, str(e)))
        dist.barrier()           # both ranks left the half-iteration together: nobody is stuck in a collective
    finally:
        dist.destroy_process_group()


def test_world2_singular_row_fails_every_rank():
    """a singular row raises on the rank that owns it only; the other rank must not walk on into the next
    half-iteration's all-reduce and block there."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_singular, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[1] == "singular row 33"
    assert "another rank reported status 1" in results[0]
