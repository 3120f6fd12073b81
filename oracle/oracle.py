"""ctypes binding of the CPU oracle (oracle/als_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py -- never from the product package (myrrix-recommender_amd), which must fail loudly when
its HIP library is missing rather than fall back to anything in here.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libals_oracle.so")

FLAG_RECONSTRUCT_R = 1
FLAG_LOSS_IGNORES_UNSPECIFIED = 2
OK = 0
SINGULAR = 1

_lib = None


def build(force=False):
    """Compile the C restatement with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "als_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libals_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_dot.restype = ctypes.c_double
        L.oracle_norm.restype = ctypes.c_double
        _lib = L
    return _lib


class SingularMatrix(Exception):
    """Mirror of SingularMatrixSolverException(apparentRank) (CMLSS:47-54)."""

    def __init__(self, row, apparent_rank):
        super().__init__("Apparent rank: %d (row %d)" % (apparent_rank, row))
        self.row = row
        self.apparent_rank = apparent_rank


def gramian(M):
    M = _f32(M)
    n, k = M.shape
    G = np.zeros((k, k), dtype=np.float64)
    lib().oracle_gramian(_ptr(M, ctypes.c_float), ctypes.c_int64(n), ctypes.c_int(k),
                         _ptr(G, ctypes.c_double))
    return G


def dot(x, y):
    x, y = _f32(x), _f32(y)
    return lib().oracle_dot(_ptr(x, ctypes.c_float), _ptr(y, ctypes.c_float), ctypes.c_int(len(x)))


def norm(x):
    x = _f32(x)
    return lib().oracle_norm(_ptr(x, ctypes.c_float), ctypes.c_int(len(x)))


def rrqr_solve(W, b, threshold=1e-5):
    W = np.ascontiguousarray(W, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    k = len(b)
    x = np.zeros(k, dtype=np.float32)
    rank = ctypes.c_int(0)
    rc = lib().oracle_rrqr_solve(_ptr(W, ctypes.c_double), _ptr(b, ctypes.c_double),
                                 ctypes.c_int(k), ctypes.c_double(threshold),
                                 _ptr(x, ctypes.c_float), ctypes.byref(rank))
    if rc == SINGULAR:
        raise SingularMatrix(-1, rank.value)
    return x


def _csr(row_ptr, col_idx, val):
    return (np.ascontiguousarray(row_ptr, dtype=np.int64),
            np.ascontiguousarray(col_idx, dtype=np.int32),
            _f32(val))


def solve_rows(row_ptr, col_idx, val, M, G, alpha=1.0, lam=0.1, flags=0, threshold=1e-5,
               row_begin=0, row_end=None, threads=1, out=None):
    """ALS:432-504 for rows [row_begin,row_end): returns the (n_rows x k) fp32 output matrix."""
    row_ptr, col_idx, val = _csr(row_ptr, col_idx, val)
    M = _f32(M)
    G = np.ascontiguousarray(G, dtype=np.float64)
    n_rows = len(row_ptr) - 1
    k = M.shape[1]
    if row_end is None:
        row_end = n_rows
    if out is None:
        out = np.zeros((n_rows, k), dtype=np.float32)
    bad_row = ctypes.c_int64(-1)
    bad_rank = ctypes.c_int(0)
    rc = lib().oracle_solve_rows(
        _ptr(row_ptr, ctypes.c_int64), _ptr(col_idx, ctypes.c_int32), _ptr(val, ctypes.c_float),
        ctypes.c_int64(row_begin), ctypes.c_int64(row_end), _ptr(M, ctypes.c_float),
        ctypes.c_int(k), _ptr(G, ctypes.c_double), ctypes.c_double(alpha), ctypes.c_double(lam),
        ctypes.c_int(flags), ctypes.c_double(threshold), _ptr(out, ctypes.c_float),
        ctypes.c_int(threads), ctypes.byref(bad_row), ctypes.byref(bad_rank))
    if rc == SINGULAR:
        raise SingularMatrix(bad_row.value, bad_rank.value)
    return out


def half_iteration(row_ptr, col_idx, val, M, alpha=1.0, lam=0.1, flags=0, threshold=1e-5,
                   threads=1):
    """ALS:340-362: G = M^T M over all rows of M, then solve every row."""
    M = _f32(M)
    return solve_rows(row_ptr, col_idx, val, M, gramian(M), alpha, lam, flags, threshold,
                      threads=threads)


def als_call(r_csr, c_csr, n_users, n_items, Y0, k, alpha=1.0, lam=0.1, flags=0,
             sing_threshold=1e-5, conv_threshold=0.001, max_iterations=30, random_y=False,
             iterate=True, test_users=None, test_items=None, threads=1):
    """ALS:176-262 call().  Returns (X, Y, iterations, convergence_value)."""
    r_row_ptr, r_col, r_val = _csr(*r_csr)
    c_row_ptr, c_col, c_val = _csr(*c_csr)
    Y = np.array(Y0, dtype=np.float32, order="C", copy=True)
    n_y = Y.shape[0]
    assert Y.shape[1] == k and n_y >= n_items
    X = np.zeros((n_users, k), dtype=np.float32)
    tu = np.arange(n_users, dtype=np.int64) if test_users is None else \
        np.ascontiguousarray(test_users, dtype=np.int64)
    ti = np.arange(n_items, dtype=np.int64) if test_items is None else \
        np.ascontiguousarray(test_items, dtype=np.int64)
    iters = ctypes.c_int(0)
    conv = ctypes.c_double(float("nan"))
    bad_row = ctypes.c_int64(-1)
    bad_rank = ctypes.c_int(0)
    rc = lib().oracle_als_call(
        _ptr(r_row_ptr, ctypes.c_int64), _ptr(r_col, ctypes.c_int32), _ptr(r_val, ctypes.c_float),
        _ptr(c_row_ptr, ctypes.c_int64), _ptr(c_col, ctypes.c_int32), _ptr(c_val, ctypes.c_float),
        ctypes.c_int64(n_users), ctypes.c_int64(n_items), ctypes.c_int64(n_y), ctypes.c_int(k),
        ctypes.c_double(alpha), ctypes.c_double(lam), ctypes.c_int(flags),
        ctypes.c_double(sing_threshold), ctypes.c_double(conv_threshold),
        ctypes.c_int(max_iterations), ctypes.c_int(1 if random_y else 0),
        ctypes.c_int(1 if iterate else 0), _ptr(tu, ctypes.c_int64), ctypes.c_int(len(tu)),
        _ptr(ti, ctypes.c_int64), ctypes.c_int(len(ti)), _ptr(X, ctypes.c_float),
        _ptr(Y, ctypes.c_float), ctypes.c_int(threads), ctypes.byref(iters), ctypes.byref(conv),
        ctypes.byref(bad_row), ctypes.byref(bad_rank))
    if rc == SINGULAR:
        raise SingularMatrix(bad_row.value, bad_rank.value)
    return X, Y, iters.value, conv.value


def multiply_xyt(X, Y):
    X, Y = _f32(X), _f32(Y)
    P = np.zeros((X.shape[0], Y.shape[0]), dtype=np.float64)
    lib().oracle_multiply_xyt(_ptr(X, ctypes.c_float), ctypes.c_int64(X.shape[0]),
                              _ptr(Y, ctypes.c_float), ctypes.c_int64(Y.shape[0]),
                              ctypes.c_int(X.shape[1]), _ptr(P, ctypes.c_double))
    return P


def dense_to_csr(R):
    """Dense matrix (0 = absent) -> CSR by row and CSR of the transpose (the RbyRow / RbyColumn
    pair MatrixUtils.addTo maintains, MU:64-71)."""
    R = np.asarray(R, dtype=np.float32)

    def one(A):
        row_ptr = [0]
        cols, vals = [], []
        for r in range(A.shape[0]):
            nz = np.nonzero(A[r])[0]
            cols.extend(nz.tolist())
            vals.extend(A[r, nz].tolist())
            row_ptr.append(len(cols))
        return (np.array(row_ptr, dtype=np.int64), np.array(cols, dtype=np.int32),
                np.array(vals, dtype=np.float32))

    return one(R), one(R.T)


def reconstruction_error(row_ptr, col_idx, val, X, Y):
    """ReconstructionEvaluator.evaluate (online/src/net/myrrix/online/eval/ReconstructionEvaluator.java:91-102):
    mean over the stored entries (u,i) of max(0, 1 - dot(X_u, Y_i)), dot = fp32 products summed in
    fp64 (SimpleVectorMath.java:34-41).  Returns (sum, count)."""
    row_ptr, col_idx, val = _csr(row_ptr, col_idx, val)
    X, Y = _f32(X), _f32(Y)
    rows = np.repeat(np.arange(len(row_ptr) - 1), np.diff(row_ptr))
    total, step = 0.0, 1 << 20
    for lo in range(0, len(col_idx), step):
        p = (X[rows[lo:lo + step]] * Y[col_idx[lo:lo + step]]).astype(np.float32)      # float * float
        d = np.cumsum(p.astype(np.float64), axis=1)[:, -1] if p.shape[1] else np.zeros(len(p))  # sequential fp64 sum
        total += float(np.sum(np.maximum(0.0, 1.0 - d)))
    return total, int(len(col_idx))
