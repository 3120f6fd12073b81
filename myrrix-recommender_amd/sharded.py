"""Row-sharded ALS across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI) for the single exchange step per half-iteration.

Sharding: within a half-iteration rows are independent given the full
opposite factor matrix and its Gramian -- exactly how the reference threads the work
(ALS:391-410, one writer per output row ALS:497-499).  Each rank holds
  * the CSR rows of its slice of users and the CSR rows (of R^T) of its slice of items,
  * FULL replicas of X and Y (torch tensors bound into the handle with mals_bind_factors),
and after solving its slice all-gathers it in place into every replica.  The shared Gramian
M^T M is computed as per-rank partials over the rank's own slice + a k x k fp64 all-reduce.
Slices are equal-sized (rows_per_rank = ceil(n/world), the tail of the last slice is zero padding
that no column index references), so the all-gather is the plain in-place NCCL all-gather.

This is the torch.distributed variant.  The product path for N > 1 is the library's own exchange below
the C-ABI (mals_group_*, group.py: cost-balanced slices, RCCL send/recv per peer); this module stays as
the multi-process alternative and as the vehicle of the world-2 gloo logic tests on CPU.

torch.distributed is plumbing here (rendezvous, RCCL communicator, stream ordering); all compute
is in libmyrrix_als.so.  `core` is duck-typed so that the world_size-2 gloo test on CPU can inject
a checker-backed stand-in (tests/test_sharded_gloo.py); the product path passes an ALSCore.
"""
from ._lib import SIDE_X, SIDE_Y


def rows_per_rank(n, world):
    return (n + world - 1) // world


class ShardedALS:
    def __init__(self, core, n_users, n_items, features, rank=0, world=1, device="cuda",
                 gramian_mode="allreduce", force_collectives=False):
        import torch
        self.torch = torch
        self.core = core
        self.k = features
        self.rank, self.world = rank, world
        self.n = {SIDE_X: n_users, SIDE_Y: n_items}
        self.per = {SIDE_X: rows_per_rank(n_users, world), SIDE_Y: rows_per_rank(n_items, world)}
        self.device = device
        self.gramian_mode = gramian_mode
        # run the collectives even with a single rank (they are no-ops then): lets one GPU exercise
        # the exact RCCL call sequence of the multi-GPU path (tests/test_gpu_sharded.py)
        self.single = world == 1 and not force_collectives
        self.F = {}
        for side in (SIDE_X, SIDE_Y):
            self.F[side] = torch.zeros(self.per[side] * world, features, dtype=torch.float32, device=device)
            core.bind_factors(side, self.F[side])
        self._gp = torch.zeros(features, features, dtype=torch.float64, device=device)
        self._bind_stream()

    def _bind_stream(self):
        """The library's kernels and torch.distributed's collectives must share one stream order: the
        handle works on torch's CURRENT stream (collectives synchronise with it, not with the null stream)."""
        if hasattr(self.core, "set_stream") and str(self.device) != "cpu" and self.torch.cuda.is_available():
            self.core.set_stream(self.torch.cuda.current_stream(self.device).cuda_stream)

    # -- data -------------------------------------------------------------------------------------
    def slice_bounds(self, side, rank=None):
        rank = self.rank if rank is None else rank
        r0 = min(self.n[side], rank * self.per[side])
        r1 = min(self.n[side], (rank + 1) * self.per[side])
        return r0, r1

    def set_matrix_from_full(self, side, row_ptr, col_idx, val):
        """Keep this rank's rows of a full CSR (numpy arrays or torch tensors)."""
        r0, r1 = self.slice_bounds(side)
        e0, e1 = int(row_ptr[r0]), int(row_ptr[r1])
        rp = row_ptr[r0:r1 + 1] - row_ptr[r0]
        self.core.set_matrix(side, rp, col_idx[e0:e1], val[e0:e1], row_offset=r0)
        self.sync_value_bound(side)

    def sync_value_bound(self, side):
        """Every rank uses the largest |value| over ALL shards for the split-precision operand
        scale, so the factors do not depend on how the rows are sharded (one scalar MAX all-reduce
        per matrix upload)."""
        if self.single:
            return
        import torch.distributed as dist
        if not hasattr(self.core, "value_stats"):   # duck-typed stand-ins of the CPU tests
            t = self.torch.tensor([self.core.value_bound(side)], dtype=self.torch.float32, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            self.core.set_value_bound(side, float(t.item()))
            return
        m, sm, n = self.core.value_stats(side)
        t = self.torch.tensor([m], dtype=self.torch.float32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q = self.torch.tensor([sm, float(n)], dtype=self.torch.float64, device=self.device)
        dist.all_reduce(q)
        self.core.set_value_stats(side, float(t.item()), float(q[0].item()) / max(float(q[1].item()), 1.0))

    def set_factors(self, side, rows):
        """Install (replicated) factor rows [0, len(rows)) -- e.g. the initial Y."""
        t = self.torch.as_tensor(rows, dtype=self.torch.float32).to(self.device)
        self.F[side][:t.shape[0]].copy_(t)

    def factors(self, side):
        return self.F[side][:self.n[side]]

    # -- one half-iteration -----------------------------------------------------------------------
    def _gramian(self, side):
        """Install G = M^T M of `side`'s factors for the next solve of the other side."""
        if self.single or self.gramian_mode == "replicated":
            self.core.gramian(side)
            return
        import torch.distributed as dist
        r0 = self.rank * self.per[side]
        self.core.gramian_partial(side, r0, self.per[side], self._gp)
        dist.all_reduce(self._gp)
        self.core.set_gramian(side, self._gp)

    def _all_gather(self, side):
        if self.single:
            return
        import torch.distributed as dist
        full = self.F[side]
        mine = full[self.rank * self.per[side]:(self.rank + 1) * self.per[side]]
        try:
            dist.all_gather_into_tensor(full, mine)
        except NotImplementedError:   # backends without the flat variant (gloo on CPU); a real RCCL failure must surface
            chunks = list(full.chunk(self.world, dim=0))
            dist.all_gather(chunks, mine.clone())

    def half_iteration(self, side):
        """iterateXFromY (ALS:340-362) for SIDE_X / iterateYFromX (ALS:367-389) for SIDE_Y."""
        self._bind_stream()
        self._gramian(1 - side)
        cr = getattr(self.core, "chunk_rows", 0)
        if self.single or cr <= 0 or cr >= self.per[side]:
            self.core.solve_side(side)
            self._all_gather(side)
            return
        # Pipelined exchange: the slice is solved in chunks of `chunk_rows` rows; as soon as a chunk
        # is solved its rows are all-gathered asynchronously (RCCL on its own stream) while the
        # next chunk is being solved.  Every rank contributes the same chunk of its (padded) slice,
        # so each collective is a plain equal-sized all-gather; padding rows are zero and stay zero.
        import torch.distributed as dist
        full, per = self.F[side], self.per[side]
        n_chunks = (per + cr - 1) // cr
        mine = self.core.num_chunks(side)
        works, stages = [], []
        k = full.shape[1]
        flat_ok = hasattr(dist, "all_gather_into_tensor") and dist.get_backend() != "gloo"
        for c in range(n_chunks):
            if c < mine:
                self.core.solve_chunk(side, c)
            lo, hi = c * cr, min((c + 1) * cr, per)
            if flat_ok:
                # one contiguous staging buffer per chunk: the flat all-gather writes it in place (no list-output
                # temporaries inside the backend), one strided copy then drops the pieces into the replica
                stage = self.torch.empty(self.world, hi - lo, k, dtype=full.dtype, device=full.device)
                works.append(dist.all_gather_into_tensor(stage.view(self.world * (hi - lo), k), full[self.rank * per + lo:self.rank * per + hi],
                                                         async_op=True))
                stages.append((stage, lo, hi))
            else:
                outs = [full[r * per + lo:r * per + hi] for r in range(self.world)]
                works.append(dist.all_gather(outs, outs[self.rank], async_op=True))
        for w in works:
            w.wait()
        for stage, lo, hi in stages:
            full.view(self.world, per, k)[:, lo:hi].copy_(stage)

    def check(self):
        """core.check() agreed over the ranks: a singular row is found by the rank that owns it only; the others
        would walk on into the next half-iteration's collectives and block there until the timeout.  Every rank
        therefore contributes its status to one MAX all-reduce and all of them raise (mals_group's agree_status
        does the same below the C-ABI)."""
        if self.single:
            self.core.check()
            return
        import torch.distributed as dist
        err = None
        try:
            self.core.check()
        except Exception as e:   # noqa: BLE001 -- re-raised below, after the collective
            err = e
        code = 0 if err is None else int(getattr(err, "status", 3) or 3)
        t = self.torch.tensor([code], dtype=self.torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if err is not None:
            raise err
        if int(t.item()) != 0:
            raise RuntimeError("another rank reported status %d in this half-iteration" % int(t.item()))

    def iterate(self, n=1, check=True):
        """check: report a singular row after EVERY half-iteration, before its zeroed factors are
        exchanged into later Gramians (the reference fails at the f.get() of that half, ALS:346-361) -- on every
        rank (self.check); check=False defers to the caller (timing loops)."""
        for _ in range(n):
            self.half_iteration(SIDE_X)
            if check:
                self.check()
            self.half_iteration(SIDE_Y)
            if check:
                self.check()
