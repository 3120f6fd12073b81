"""ALSCore: thin object wrapper over one mals_handle (one GPU).  Pure plumbing: every method is a
single C-ABI call (include/myrrix_als.h); arrays are numpy (host) or torch CUDA tensors (device,
borrowed -- a reference is kept so they stay alive)."""
import ctypes

import numpy as np

from . import _lib
from ._lib import (FLAG_LOSS_IGNORES_UNSPECIFIED, FLAG_RECONSTRUCT_R, MEM_DEVICE, MEM_HOST, SIDE_X,
                   SIDE_Y)


class MalsError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (_lib.STATUS_NAMES.get(status, status), message))
        self.status = status
        self.message = message


class SingularSystem(MalsError):
    """MALS_SINGULAR: the reference throws SingularMatrixSolverException here (CMLSS:46-54)."""

    def __init__(self, status, message, side, row, apparent_rank):
        super().__init__(status, message)
        self.side = side
        self.row = row
        self.apparent_rank = apparent_rank


class Cancelled(MalsError):
    pass


class IllConditioned(MalsError):
    """MALS_ILL_CONDITIONED: IllConditionedSolverException (Generation.java:150-153)."""

    def __init__(self, status, message, inf_norm):
        super().__init__(status, message)
        self.inf_norm = inf_norm


class HostSolver:
    """One mals_solver: the fp64 pivoted-QR factorization of a k x k matrix (Solver.java:35-41)."""

    def __init__(self, L, handle):
        self._L, self._s = L, handle
        self.n = L.mals_solver_dim(handle)

    @classmethod
    def create(cls, A, singularity_threshold=1e-5):
        """MatrixUtils.getSolver(A) (MU:137, CMLSS:37-55); raises SingularSystem with the rank."""
        L = _lib.load()
        A = _host(A, np.float64)
        assert A.ndim == 2 and A.shape[0] == A.shape[1]
        s, rank = ctypes.c_void_p(), ctypes.c_int32(0)
        rc = L.mals_solver_create(A.ctypes.data_as(ctypes.c_void_p), A.shape[0], float(singularity_threshold),
                                  ctypes.byref(s), ctypes.byref(rank))
        if rc == _lib.SINGULAR:
            raise SingularSystem(rc, "Apparent rank: %d" % rank.value, -1, -1, rank.value)
        if rc != _lib.OK:
            raise MalsError(rc, "mals_solver_create failed")
        return cls(L, s)

    def solve_dtof(self, b):
        b = _host(b, np.float64)
        assert b.shape == (self.n,)
        x = np.empty(self.n, dtype=np.float32)
        rc = self._L.mals_solver_solve_dtof(self._s, b.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p))
        if rc != _lib.OK:
            raise MalsError(rc, "mals_solver_solve_dtof failed")
        return x

    def solve_ftod(self, b):
        b = _host(b, np.float32)
        assert b.shape == (self.n,)
        x = np.empty(self.n, dtype=np.float64)
        rc = self._L.mals_solver_solve_ftod(self._s, b.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p))
        if rc != _lib.OK:
            raise MalsError(rc, "mals_solver_solve_ftod failed")
        return x

    def close(self):
        if getattr(self, "_s", None) is not None and self._s.value:
            self._L.mals_solver_destroy(self._s)
            self._s = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _host(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class ALSCore:
    def __init__(self, features, alpha=1.0, lam=0.1, flags=0, device=0, segment_nnz=0,
                 singularity_threshold=1e-5, chunk_rows=0, gramian_mode=0, solve_mode=0):
        self._L = _lib.load()
        cfg = _lib.Config()
        self._L.mals_default_config(ctypes.byref(cfg))
        cfg.features = int(features)
        cfg.alpha = float(alpha)
        cfg.lam = float(lam)
        cfg.flags = int(flags)
        cfg.device = int(device)
        cfg.segment_nnz = int(segment_nnz)
        cfg.chunk_rows = int(chunk_rows)
        cfg.gramian_mode = int(gramian_mode)
        cfg.solve_mode = int(solve_mode)
        self.chunk_rows = max(0, int(chunk_rows))
        cfg.singularity_threshold = float(singularity_threshold)
        self.features = int(features)
        self._h = ctypes.c_void_p()
        self._keep = {}
        rc = self._L.mals_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc != _lib.OK:
            self._h = ctypes.c_void_p()
            raise MalsError(rc, "mals_create failed (features=%d, device=%d): a HIP device is "
                                "required, there is no CPU fallback" % (features, device))

    # -- lifecycle --------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.mals_destroy(self._h)
            self._h = ctypes.c_void_p()
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc == _lib.OK:
            return
        msg = (self._L.mals_last_error(self._h) or b"").decode("utf-8", "replace")
        if rc == _lib.SINGULAR:
            side, row, rank = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int32()
            self._L.mals_singular_info(self._h, ctypes.byref(side), ctypes.byref(row), ctypes.byref(rank))
            raise SingularSystem(rc, msg, side.value, row.value, rank.value)
        if rc == _lib.CANCELLED:
            raise Cancelled(rc, msg)
        raise MalsError(rc, msg)

    # -- configuration ----------------------------------------------------------------------------
    def set_stream(self, hip_stream_ptr):
        self._chk(self._L.mals_set_stream(self._h, ctypes.c_void_p(int(hip_stream_ptr or 0))))

    def set_factor_rows(self, side, n_rows_total):
        self._keep.pop(("F", side), None)
        self._chk(self._L.mals_set_factor_rows(self._h, side, int(n_rows_total)))

    def bind_factors(self, side, tensor):
        """tensor: contiguous fp32 torch CUDA tensor [n_rows_total, features]."""
        assert _is_torch(tensor) and tensor.is_cuda and tensor.is_contiguous()
        assert tensor.dim() == 2 and tensor.shape[1] == self.features and str(tensor.dtype) == "torch.float32"
        self._keep[("F", side)] = tensor
        self._chk(self._L.mals_bind_factors(self._h, side, ctypes.c_void_p(tensor.data_ptr()),
                                            int(tensor.shape[0])))

    def factor_device_ptr(self, side):
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        self._chk(self._L.mals_factor_device_ptr(self._h, side, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def set_matrix(self, side, row_ptr, col_idx, val, row_offset=0):
        if _is_torch(row_ptr):
            assert row_ptr.is_cuda and col_idx.is_cuda and val.is_cuda
            assert str(row_ptr.dtype) == "torch.int64" and str(col_idx.dtype) == "torch.int32" \
                and str(val.dtype) == "torch.float32"
            row_ptr, col_idx, val = row_ptr.contiguous(), col_idx.contiguous(), val.contiguous()
            self._keep[("M", side)] = (row_ptr, col_idx, val)
            n_rows, nnz = int(row_ptr.shape[0]) - 1, int(col_idx.shape[0])
            self._chk(self._L.mals_set_matrix(
                self._h, side, int(row_offset), n_rows, nnz, ctypes.c_void_p(row_ptr.data_ptr()),
                ctypes.c_void_p(col_idx.data_ptr()), ctypes.c_void_p(val.data_ptr()), MEM_DEVICE))
        else:
            row_ptr = _host(row_ptr, np.int64)
            col_idx = _host(col_idx, np.int32)
            val = _host(val, np.float32)
            self._keep.pop(("M", side), None)
            n_rows, nnz = len(row_ptr) - 1, len(col_idx)
            self._chk(self._L.mals_set_matrix(
                self._h, side, int(row_offset), n_rows, nnz, row_ptr.ctypes.data_as(ctypes.c_void_p),
                col_idx.ctypes.data_as(ctypes.c_void_p), val.ctypes.data_as(ctypes.c_void_p), MEM_HOST))

    def set_matrix_chunked(self, side, row_ptr, col_idx, val, rows_per_chunk, row_offset=0):
        """Same as set_matrix (host arrays) through the begin/append/end entry points."""
        row_ptr = _host(row_ptr, np.int64)
        col_idx = _host(col_idx, np.int32)
        val = _host(val, np.float32)
        n_rows, nnz = len(row_ptr) - 1, len(col_idx)
        self._chk(self._L.mals_begin_matrix(self._h, side, int(row_offset), n_rows, nnz))
        for r0 in range(0, max(n_rows, 1), rows_per_chunk):
            r1 = min(n_rows, r0 + rows_per_chunk)
            rp = np.ascontiguousarray(row_ptr[r0:r1 + 1] - row_ptr[r0])
            c = np.ascontiguousarray(col_idx[row_ptr[r0]:row_ptr[r1]])
            v = np.ascontiguousarray(val[row_ptr[r0]:row_ptr[r1]])
            self._chk(self._L.mals_append_rows(
                self._h, side, r1 - r0, rp.ctypes.data_as(ctypes.c_void_p),
                c.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p)))
        self._chk(self._L.mals_end_matrix(self._h, side))

    def value_bound(self, side):
        v = ctypes.c_float(0.0)
        self._chk(self._L.mals_get_value_bound(self._h, side, ctypes.byref(v)))
        return v.value

    def set_value_bound(self, side, max_abs_value):
        self._chk(self._L.mals_set_value_bound(self._h, side, float(max_abs_value)))

    def value_stats(self, side):
        """(largest |value|, sum of |value|, number of entries) of the local rows."""
        m, sm, n = ctypes.c_float(0.0), ctypes.c_double(0.0), ctypes.c_int64(0)
        self._chk(self._L.mals_get_value_stats(self._h, side, ctypes.byref(m), ctypes.byref(sm), ctypes.byref(n)))
        return m.value, sm.value, n.value

    def set_value_stats(self, side, max_abs_value, mean_abs_value):
        self._chk(self._L.mals_set_value_stats(self._h, side, float(max_abs_value), float(mean_abs_value)))

    # -- factors ----------------------------------------------------------------------------------
    def set_factors(self, side, rows, row_begin=0):
        rows = _host(rows, np.float32)
        assert rows.ndim == 2 and rows.shape[1] == self.features
        self._chk(self._L.mals_set_factors(self._h, side, int(row_begin), rows.shape[0],
                                           rows.ctypes.data_as(ctypes.c_void_p)))

    def get_factors(self, side, row_begin=0, n_rows=None):
        if n_rows is None:
            n_rows = self.factor_device_ptr(side)[1] - row_begin
        out = np.empty((n_rows, self.features), dtype=np.float32)
        self._chk(self._L.mals_get_factors(self._h, side, int(row_begin), int(n_rows),
                                           out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def get_rows(self, side, idx):
        idx = _host(idx, np.int64)
        out = np.empty((len(idx), self.features), dtype=np.float32)
        self._chk(self._L.mals_get_rows(self._h, side, idx.ctypes.data_as(ctypes.c_void_p), len(idx),
                                        out.ctypes.data_as(ctypes.c_void_p)))
        return out

    # -- compute ----------------------------------------------------------------------------------
    def gramian(self, side, fetch=False):
        if fetch:
            G = np.empty((self.features, self.features), dtype=np.float64)
            self._chk(self._L.mals_gramian(self._h, side, G.ctypes.data_as(ctypes.c_void_p)))
            return G
        self._chk(self._L.mals_gramian(self._h, side, None))
        return None

    def gramian_partial(self, side, row_begin, n_rows, out_tensor):
        assert _is_torch(out_tensor) and out_tensor.is_cuda and str(out_tensor.dtype) == "torch.float64"
        assert out_tensor.numel() == self.features * self.features and out_tensor.is_contiguous()
        self._chk(self._L.mals_gramian_partial(self._h, side, int(row_begin), int(n_rows),
                                               ctypes.c_void_p(out_tensor.data_ptr())))

    def set_gramian(self, side, G):
        if _is_torch(G):
            assert G.is_cuda and str(G.dtype) == "torch.float64" and G.is_contiguous()
            self._chk(self._L.mals_set_gramian(self._h, side, ctypes.c_void_p(G.data_ptr()), MEM_DEVICE))
        else:
            G = _host(G, np.float64)
            self._chk(self._L.mals_set_gramian(self._h, side, G.ctypes.data_as(ctypes.c_void_p), MEM_HOST))

    def solve_side(self, side):
        self._chk(self._L.mals_solve_side(self._h, side))

    def solve_chunk(self, side, chunk):
        self._chk(self._L.mals_solve_chunk(self._h, side, int(chunk)))

    def num_chunks(self, side):
        n = ctypes.c_int32()
        self._chk(self._L.mals_num_chunks(self._h, side, ctypes.byref(n)))
        return n.value

    def check(self):
        self._chk(self._L.mals_check(self._h))

    def half_iteration(self, side):
        self._chk(self._L.mals_half_iteration(self._h, side))

    def factorize(self, convergence_threshold, max_iterations, random_y, test_users, test_items,
                  iterate=True):
        tu = _host(test_users, np.int64)
        ti = _host(test_items, np.int64)
        iters, conv = ctypes.c_int32(0), ctypes.c_double(float("nan"))
        rc = self._L.mals_factorize(self._h, float(convergence_threshold), int(max_iterations),
                                    1 if random_y else 0, 1 if iterate else 0,
                                    tu.ctypes.data_as(ctypes.c_void_p), len(tu),
                                    ti.ctypes.data_as(ctypes.c_void_p), len(ti),
                                    ctypes.byref(iters), ctypes.byref(conv))
        self._chk(rc)
        return iters.value, conv.value

    def set_iteration_callback(self, fn):
        """mals_set_iteration_callback: fn(info: dict) after every iteration of factorize() -- iteration, avg_abs_difference,
        seconds, x_rows, y_rows, entries_gathered, algorithmic_bytes, devices (what call() logs per iteration, ALS:241-246,
        351-358); None removes it."""
        if fn is None:
            self._iter_cb = None
            self._chk(self._L.mals_set_iteration_callback(self._h, None, None))
            return

        def tramp(_user, info):
            i = info.contents
            fn({name: getattr(i, name) for name, _ in _lib.IterationInfo._fields_ if name not in ("struct_size", "reserved")})
        self._iter_cb = _lib.ITERATION_FN(tramp)      # keep the trampoline alive as long as the library may call it
        self._chk(self._L.mals_set_iteration_callback(self._h, ctypes.cast(self._iter_cb, ctypes.c_void_p), None))

    def cancel(self):
        self._chk(self._L.mals_cancel(self._h))

    def recommend(self, user_idx, how_many, consider_known_items=False):
        """ServerRecommender.recommend for model users (dense indices): (item_idx [q][how_many] int64,
        scores float32, counts)."""
        u = _host(user_idx, np.int64)
        idx = np.empty((len(u), how_many), dtype=np.int64)
        sc = np.empty((len(u), how_many), dtype=np.float32)
        cnt = np.empty(len(u), dtype=np.int32)
        self._chk(self._L.mals_recommend(self._h, u.ctypes.data_as(ctypes.c_void_p), len(u), int(how_many),
                                         1 if consider_known_items else 0, idx.ctypes.data_as(ctypes.c_void_p),
                                         sc.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)))
        return idx, sc, cnt

    def recommend_vectors(self, vectors, how_many, exclude=None):
        """Top-N for caller-supplied query vectors; exclude: optional list of item-index lists."""
        v = _host(vectors, np.float32)
        assert v.ndim == 2 and v.shape[1] == self.features
        idx = np.empty((len(v), how_many), dtype=np.int64)
        sc = np.empty((len(v), how_many), dtype=np.float32)
        cnt = np.empty(len(v), dtype=np.int32)
        ep = ei = None
        if exclude is not None:
            ptr = np.zeros(len(v) + 1, dtype=np.int64)
            ptr[1:] = np.cumsum([len(e) for e in exclude])
            flat = np.ascontiguousarray(np.concatenate([np.asarray(e, np.int64) for e in exclude]) if ptr[-1] else np.zeros(0, np.int64))
            ep, ei = ptr.ctypes.data_as(ctypes.c_void_p), flat.ctypes.data_as(ctypes.c_void_p)
            self._keep[("excl",)] = (ptr, flat)
        self._chk(self._L.mals_recommend_vectors(self._h, v.ctypes.data_as(ctypes.c_void_p), len(v), int(how_many), ep, ei,
                                                 idx.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p),
                                                 cnt.ctypes.data_as(ctypes.c_void_p)))
        return idx, sc, cnt

    def set_known_items(self, row_ptr, item_idx):
        """knownItemIDs as a CSR over the local user rows (generation.getKnownItemIDs()); None, None: back to the rows of R."""
        if row_ptr is None:
            self._chk(self._L.mals_set_known_items(self._h, 0, None, None, MEM_HOST))
            return
        rp = _host(row_ptr, np.int64)
        ii = _host(item_idx, np.int32)
        self._chk(self._L.mals_set_known_items(self._h, len(rp) - 1, rp.ctypes.data_as(ctypes.c_void_p), ii.ctypes.data_as(ctypes.c_void_p), MEM_HOST))

    def set_tag_items(self, item_idx):
        """userTagIDs as dense item indices: rows of Y that no recommend call returns (RecommendIterator.java:72); None or
        empty clears."""
        ii = _host(item_idx if item_idx is not None else [], np.int64)
        self._chk(self._L.mals_set_tag_items(self._h, len(ii), ii.ctypes.data_as(ctypes.c_void_p) if len(ii) else None, MEM_HOST))

    def tag_item_count(self):
        n = ctypes.c_int64(0)
        self._chk(self._L.mals_get_tag_item_count(self._h, ctypes.byref(n)))
        return n.value

    def recommend_front_stats(self):
        """{calls, queries, passes, exclusive} of the serving front since the handle was created."""
        o = np.zeros(4, dtype=np.int64)
        self._chk(self._L.mals_recommend_front_stats(self._h, o.ctypes.data_as(ctypes.c_void_p)))
        return {"calls": int(o[0]), "queries": int(o[1]), "passes": int(o[2]), "exclusive": int(o[3])}

    def recommend_set_depth(self, passes_in_flight):
        self._chk(self._L.mals_recommend_set_depth(self._h, int(passes_in_flight)))

    def recommend_set_spin_us(self, spin_us):
        self._chk(self._L.mals_recommend_set_spin_us(self._h, int(spin_us)))

    def recommend_to_many(self, queries, how_many, exclude=None):
        """recommendToMany (ServerRecommender.java:366-441): queries = list of (n_j x features) arrays, one per query; the
        score of an item is the mean of its dots with the query's vectors (RecommendIterator.java:93-104)."""
        qs = [np.ascontiguousarray(np.atleast_2d(_host(q, np.float32))) for q in queries]
        assert all(q.shape[1] == self.features for q in qs)
        vptr = np.zeros(len(qs) + 1, dtype=np.int64)
        vptr[1:] = np.cumsum([len(q) for q in qs])
        flatv = np.ascontiguousarray(np.concatenate(qs, axis=0)) if qs else np.zeros((0, self.features), np.float32)
        idx = np.empty((len(qs), how_many), dtype=np.int64)
        sc = np.empty((len(qs), how_many), dtype=np.float32)
        cnt = np.empty(len(qs), dtype=np.int32)
        ep = ei = None
        if exclude is not None:
            ptr = np.zeros(len(qs) + 1, dtype=np.int64)
            ptr[1:] = np.cumsum([len(e) for e in exclude])
            flat = np.ascontiguousarray(np.concatenate([np.asarray(e, np.int64) for e in exclude]) if ptr[-1] else np.zeros(0, np.int64))
            ep, ei = ptr.ctypes.data_as(ctypes.c_void_p), flat.ctypes.data_as(ctypes.c_void_p)
            self._keep[("excl",)] = (ptr, flat)
        self._chk(self._L.mals_recommend_to_many(self._h, flatv.ctypes.data_as(ctypes.c_void_p), vptr.ctypes.data_as(ctypes.c_void_p),
                                                 len(qs), int(how_many), ep, ei, idx.ctypes.data_as(ctypes.c_void_p),
                                                 sc.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)))
        return idx, sc, cnt

    def reconstruction_error(self):
        """ReconstructionEvaluator's sum and count over the local user rows (mean = sum / count)."""
        sm, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
        self._chk(self._L.mals_reconstruction_error(self._h, ctypes.byref(sm), ctypes.byref(cnt)))
        return sm.value, cnt.value

    def recompute_solver(self, side):
        """Generation.recomputeSolver (Generation.java:142-158) for `side`'s factors.  Returns
        (HostSolver or None for an empty side, norm)."""
        s, norm = ctypes.c_void_p(), ctypes.c_double(0.0)
        rc = self._L.mals_recompute_solver(self._h, side, ctypes.byref(s), ctypes.byref(norm))
        if rc == _lib.ILL_CONDITIONED:
            msg = (self._L.mals_last_error(self._h) or b"").decode("utf-8", "replace")
            raise IllConditioned(rc, msg, norm.value)
        self._chk(rc)
        return (HostSolver(self._L, s) if s.value else None), norm.value

    # -- stats ------------------------------------------------------------------------------------
    def enable_timing(self, on=True):
        self._chk(self._L.mals_enable_timing(self._h, 1 if on else 0))

    def reset_stats(self):
        self._chk(self._L.mals_reset_stats(self._h))

    def sample_dots(self, test_users, test_items):
        """est[i, j] = SimpleVectorMath.dot(X[test_users[i]], Y[test_items[j]]) from the resident factors."""
        tu = _host(test_users, np.int64)
        ti = _host(test_items, np.int64)
        out = np.empty((len(tu), len(ti)), dtype=np.float64)
        self._chk(self._L.mals_sample_dots(self._h, tu.ctypes.data_as(ctypes.c_void_p), len(tu), ti.ctypes.data_as(ctypes.c_void_p),
                                           len(ti), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def timeline(self):
        """mals_get_timeline: host microseconds [first kernel of the half-iteration enqueued, eigendecomposition begin, end, -]."""
        out = (ctypes.c_double * 4)()
        self._chk(self._L.mals_get_timeline(self._h, out))
        return list(out)

    def gather_scale(self):
        """mals_get_gather_scale: [S, 1/S^2, range flag, bound on |y| used] of the last half-iteration's split-precision gather."""
        out = (ctypes.c_float * 4)()
        self._chk(self._L.mals_get_gather_scale(self._h, out))
        return list(out)

    def set_refine_limit(self, limit):
        """mals_set_refine_limit: conditioning estimate above which a row is re-solved with fp64 residuals (0 = never)."""
        self._chk(self._L.mals_set_refine_limit(self._h, float(limit)))

    def stats(self):
        st = _lib.Stats()
        self._chk(self._L.mals_get_stats(self._h, ctypes.byref(st)))
        return {name: getattr(st, name) for name, _ in _lib.Stats._fields_ if name not in ("struct_size", "reserved")}


__all__ = ["ALSCore", "HostSolver", "MalsError", "SingularSystem", "Cancelled", "IllConditioned", "SIDE_X", "SIDE_Y",
           "FLAG_RECONSTRUCT_R", "FLAG_LOSS_IGNORES_UNSPECIFIED"]
