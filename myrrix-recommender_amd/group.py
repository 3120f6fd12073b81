"""ctypes binding of the mals_group_* entry points: the multi-GPU half-iteration below the C-ABI
(csrc/mals_group.cpp).  Two constructors:

  GroupALS.single_process(features, devices, ...)   one process drives all GPUs (ncclCommInitAll, or
                                                    peer copies with backend=GROUP_PEER_COPY)
  GroupALS.from_torch_distributed(features, ...)    one process per GPU: the RCCL unique id is broadcast
                                                    with torch.distributed (plumbing: rendezvous only);
                                                    every collective afterwards is the library's own

Every method is a single C-ABI call; in multi-process groups the calls are collective.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import MEM_DEVICE, MEM_HOST, SIDE_X, SIDE_Y
from .core import ALSCore, Cancelled, MalsError, SingularSystem


def plan_shards(row_ptr, world, features, row_cost=-1.0):
    """mals_plan_shards: contiguous slices of equal cost (entries + row_cost per row).  Host only."""
    L = _lib.load()
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    out = np.zeros(world + 1, dtype=np.int64)
    rc = L.mals_plan_shards(rp.ctypes.data_as(ctypes.c_void_p), len(rp) - 1, int(world), float(row_cost), int(features),
                            out.ctypes.data_as(ctypes.c_void_p))
    if rc != _lib.OK:
        raise MalsError(rc, "mals_plan_shards")
    return out


class _MemberCore(ALSCore):
    """ALSCore view of a group member's handle (owned by the group: never destroyed here)."""

    def __init__(self, L, handle, features):   # noqa: D107 -- no mals_create
        self._L = L
        self._h = handle
        self._keep = {}
        self.features = features
        self.chunk_rows = 0

    def close(self):
        self._h = ctypes.c_void_p()

    def __del__(self):
        pass


class GroupALS:
    def __init__(self, features, handle, world):
        self._L = _lib.load()
        self._g = handle
        self.features = int(features)
        self.world = int(world)
        self._keep = {}

    @staticmethod
    def _config(L, features, alpha, lam, flags, device, segment_nnz, singularity_threshold, gramian_mode, solve_mode):
        cfg = _lib.Config()
        L.mals_default_config(ctypes.byref(cfg))
        cfg.features, cfg.alpha, cfg.lam, cfg.flags = int(features), float(alpha), float(lam), int(flags)
        cfg.device, cfg.segment_nnz = int(device), int(segment_nnz)
        cfg.singularity_threshold = float(singularity_threshold)
        cfg.gramian_mode, cfg.solve_mode = int(gramian_mode), int(solve_mode)
        return cfg

    @classmethod
    def single_process(cls, features, devices, backend=_lib.GROUP_RCCL, alpha=1.0, lam=0.1, flags=0, segment_nnz=0,
                       singularity_threshold=1e-5, gramian_mode=0, solve_mode=0, exchange_chunks=4):
        L = _lib.load()
        cfg = cls._config(L, features, alpha, lam, flags, devices[0], segment_nnz, singularity_threshold, gramian_mode, solve_mode)
        devs = (ctypes.c_int32 * len(devices))(*devices)
        g = ctypes.c_void_p()
        rc = L.mals_group_create(ctypes.byref(cfg), devs, len(devices), int(backend), ctypes.byref(g))
        if rc != _lib.OK:
            raise MalsError(rc, "mals_group_create failed (devices %s): %s" % (list(devices), _lib.create_error()))
        self = cls(features, g, len(devices))
        self._chk(L.mals_group_set_exchange_chunks(g, int(exchange_chunks)))
        return self

    def set_alternate_streams(self, on):
        """mals_group_set_alternate_streams: consecutive chunks of a half-iteration on two alternating compute streams
        (default on) or on one (A/B)."""
        self._chk(self._L.mals_group_set_alternate_streams(self._g, 1 if on else 0))

    @staticmethod
    def use_transport(library_path):
        """mals_group_use_transport: the shared library to load as RCCL (None = librccl.so.1), once per process and
        before the first group / unique id.  The tests point it at their stand-in transport."""
        L = _lib.load()
        rc = L.mals_group_use_transport(library_path.encode() if library_path else None)
        if rc != _lib.OK:
            raise MalsError(rc, "mals_group_use_transport: a transport is already loaded in this process")

    def comm_info(self, i=0):
        """What the communicator reports for local member i: {"comm_size", "comm_rank", "hip_device", "pci_bus_id"}."""
        n, r, d = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        buf = ctypes.create_string_buffer(32)
        self._chk(self._L.mals_group_comm_info(self._g, int(i), ctypes.byref(n), ctypes.byref(r), ctypes.byref(d), buf, 32))
        return {"comm_size": n.value, "comm_rank": r.value, "hip_device": d.value, "pci_bus_id": buf.value.decode("ascii", "replace")}

    @staticmethod
    def unique_id():
        """mals_group_unique_id: the 128-byte RCCL unique id rank 0 creates and hands to the other ranks."""
        L = _lib.load()
        buf = (ctypes.c_uint8 * 128)()
        rc = L.mals_group_unique_id(buf)
        if rc != _lib.OK:
            raise MalsError(rc, "mals_group_unique_id failed (is librccl.so.1 loadable?)")
        return bytes(buf)

    @classmethod
    def from_unique_id(cls, features, device, world, rank, uid, alpha=1.0, lam=0.1, flags=0, segment_nnz=0,
                       singularity_threshold=1e-5, gramian_mode=0, solve_mode=0, exchange_chunks=4):
        """One rank per process (mals_group_create_rank); `uid` = unique_id() of rank 0, carried by whatever
        the application has (MPI, a socket, torch.distributed); None with world == 1 = no communicator at all."""
        L = _lib.load()
        cfg = cls._config(L, features, alpha, lam, flags, device, segment_nnz, singularity_threshold, gramian_mode, solve_mode)
        buf = (ctypes.c_uint8 * 128)(*uid) if uid is not None else None
        g = ctypes.c_void_p()
        rc = L.mals_group_create_rank(ctypes.byref(cfg), int(world), int(rank), buf, ctypes.byref(g))
        if rc != _lib.OK:
            raise MalsError(rc, "mals_group_create_rank failed (rank %d of %d): %s" % (rank, world, _lib.create_error()))
        self = cls(features, g, world)
        self._chk(L.mals_group_set_exchange_chunks(g, int(exchange_chunks)))
        return self

    @classmethod
    def from_torch_distributed(cls, features, device, alpha=1.0, lam=0.1, flags=0, segment_nnz=0, singularity_threshold=1e-5,
                               gramian_mode=0, solve_mode=0, exchange_chunks=4, world=None, rank=None, one_rank_communicator=False):
        """One rank per process.  torch.distributed (already initialised unless world == 1) only carries
        the 128-byte RCCL unique id from rank 0 to the others."""
        if world is None:
            import torch.distributed as dist
            world, rank = dist.get_world_size(), dist.get_rank()
        uid = None
        if world > 1 or one_rank_communicator:
            uid = cls.unique_id() if rank == 0 else bytes(128)
            if world > 1:
                import torch
                import torch.distributed as dist
                # a CPU tensor under gloo (single-device test runs), a device tensor under RCCL
                dev = torch.device("cuda", device) if dist.get_backend() == "nccl" else torch.device("cpu")
                t = torch.tensor(list(uid), dtype=torch.uint8, device=dev)
                dist.broadcast(t, 0)
                uid = bytes(t.cpu().tolist())
        return cls.from_unique_id(features, device, world, rank, uid, alpha=alpha, lam=lam, flags=flags, segment_nnz=segment_nnz,
                                  singularity_threshold=singularity_threshold, gramian_mode=gramian_mode, solve_mode=solve_mode,
                                  exchange_chunks=exchange_chunks)

    # -- lifecycle --------------------------------------------------------------------------------
    def close(self):
        if self._g is not None and self._g.value:
            self._L.mals_group_destroy(self._g)
            self._g = ctypes.c_void_p()
            self._keep.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc == _lib.OK:
            return
        msg = (self._L.mals_group_last_error(self._g) or b"").decode("utf-8", "replace")
        if rc == _lib.SINGULAR:
            side, row, rank = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int32()
            self._L.mals_group_singular_info(self._g, ctypes.byref(side), ctypes.byref(row), ctypes.byref(rank))
            raise SingularSystem(rc, msg, side.value, row.value, rank.value)
        if rc == _lib.CANCELLED:
            raise Cancelled(rc, msg)
        raise MalsError(rc, msg)

    def set_refine_limit(self, limit):
        """mals_group_set_refine_limit: see mals_set_refine_limit (every local member)."""
        self._chk(self._L.mals_group_set_refine_limit(self._g, float(limit)))

    def local(self, i=0):
        """(ALSCore view, rank) of local member i -- stats, top-N, reconstruction error per GPU."""
        h, r = ctypes.c_void_p(), ctypes.c_int32()
        self._chk(self._L.mals_group_local(self._g, int(i), ctypes.byref(h), ctypes.byref(r)))
        return _MemberCore(self._L, h, self.features), r.value

    # -- data -------------------------------------------------------------------------------------
    def set_factor_rows(self, side, n_rows_total):
        self._chk(self._L.mals_group_set_factor_rows(self._g, side, int(n_rows_total)))

    def set_factors(self, side, rows, row_begin=0):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        self._chk(self._L.mals_group_set_factors(self._g, side, int(row_begin), rows.shape[0], rows.ctypes.data_as(ctypes.c_void_p)))

    def get_factors(self, side, row_begin, n_rows):
        out = np.empty((n_rows, self.features), dtype=np.float32)
        self._chk(self._L.mals_group_get_factors(self._g, side, int(row_begin), int(n_rows), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def set_matrix(self, side, row_ptr, col_idx, val):
        """The FULL matrix of a side (numpy arrays, or torch CUDA tensors of this process's device)."""
        if type(row_ptr).__module__.startswith("torch"):
            row_ptr, col_idx, val = row_ptr.contiguous(), col_idx.contiguous(), val.contiguous()
            self._keep[("M", side)] = (row_ptr, col_idx, val)
            self._chk(self._L.mals_group_set_matrix(self._g, side, int(row_ptr.shape[0]) - 1, int(col_idx.shape[0]),
                                                    ctypes.c_void_p(row_ptr.data_ptr()), ctypes.c_void_p(col_idx.data_ptr()),
                                                    ctypes.c_void_p(val.data_ptr()), MEM_DEVICE))
        else:
            rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
            c = np.ascontiguousarray(col_idx, dtype=np.int32)
            v = np.ascontiguousarray(val, dtype=np.float32)
            self._chk(self._L.mals_group_set_matrix(self._g, side, len(rp) - 1, len(c), rp.ctypes.data_as(ctypes.c_void_p),
                                                    c.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p), MEM_HOST))

    def set_matrix_chunked(self, side, row_ptr, col_idx, val, rows_per_piece):
        rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
        c = np.ascontiguousarray(col_idx, dtype=np.int32)
        v = np.ascontiguousarray(val, dtype=np.float32)
        n = len(rp) - 1
        self._chk(self._L.mals_group_begin_matrix(self._g, side, n, rp.ctypes.data_as(ctypes.c_void_p)))
        for r0 in range(0, n, rows_per_piece):
            r1 = min(n, r0 + rows_per_piece)
            cc = np.ascontiguousarray(c[rp[r0]:rp[r1]])
            vv = np.ascontiguousarray(v[rp[r0]:rp[r1]])
            self._chk(self._L.mals_group_append_rows(self._g, side, r1 - r0, cc.ctypes.data_as(ctypes.c_void_p), vv.ctypes.data_as(ctypes.c_void_p)))
        self._chk(self._L.mals_group_end_matrix(self._g, side))

    def recommend(self, user_idx, how_many, consider_known_items=False):
        """ServerRecommender.recommend on the group: every query answered by the member that holds the user's row."""
        u = np.ascontiguousarray(user_idx, dtype=np.int64)
        idx = np.empty((len(u), how_many), dtype=np.int64)
        sc = np.empty((len(u), how_many), dtype=np.float32)
        cnt = np.empty(len(u), dtype=np.int32)
        self._chk(self._L.mals_group_recommend(self._g, u.ctypes.data_as(ctypes.c_void_p), len(u), int(how_many), 1 if consider_known_items else 0,
                                               idx.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)))
        return idx, sc, cnt

    def bounds(self, side):
        out = np.zeros(self.world + 1, dtype=np.int64)
        self._chk(self._L.mals_group_bounds(self._g, side, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    # -- compute ----------------------------------------------------------------------------------
    def half_iteration(self, side):
        self._chk(self._L.mals_group_half_iteration(self._g, side))

    def iterate(self, n=1):
        for _ in range(n):
            self.half_iteration(SIDE_X)
            self.half_iteration(SIDE_Y)

    def exchange_only(self, side):
        self._chk(self._L.mals_group_exchange_only(self._g, side))

    def synchronize(self):
        self._chk(self._L.mals_group_synchronize(self._g))

    def factorize(self, convergence_threshold, max_iterations, random_y, iterate, test_users, test_items):
        tu = np.ascontiguousarray(test_users, dtype=np.int64)
        ti = np.ascontiguousarray(test_items, dtype=np.int64)
        it, conv = ctypes.c_int32(0), ctypes.c_double(float("nan"))
        self._chk(self._L.mals_group_factorize(self._g, float(convergence_threshold), int(max_iterations), int(bool(random_y)),
                                               int(bool(iterate)), tu.ctypes.data_as(ctypes.c_void_p), len(tu),
                                               ti.ctypes.data_as(ctypes.c_void_p), len(ti), ctypes.byref(it), ctypes.byref(conv)))
        return it.value, conv.value


__all__ = ["GroupALS", "plan_shards", "SIDE_X", "SIDE_Y"]
