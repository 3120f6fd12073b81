"""numpy restatement of what als_refine_kernel does to a marked row (csrc/als_kernels.h), on the CPU, against the
oracle: the fp32 system and its fp32 Cholesky factor as preconditioner, conjugate gradients on the exact fp64 system;
and of the mark itself (max(largest entry of the row's own part, a quarter of the largest entry of W) over the
smallest pivot).  Case 103 of the seeded sweep (tests/test_gpu_fuzz.py): alpha = 40 on values up to 150 against
lambda = 0.01 -- cond(W) ~ 5e4, where fp32 alone is 5e-4 off the reference."""
import numpy as np

from oracle import oracle
from tests.test_gpu_fuzz import draw_case


def row_system(csr, M, G, r, alpha, lam, dtype):
    rp, col, val = csr
    a, b = rp[r], rp[r + 1]
    y = M[col[a:b]].astype(dtype)
    v = val[a:b].astype(dtype)
    w = dtype(alpha) * np.abs(v)
    cb = np.where(v > 0, dtype(1) + w, dtype(0)).astype(dtype)
    part = ((y.T * w) @ y).astype(dtype)
    W = (G.astype(dtype) + part + dtype(lam * alpha * (b - a)) * np.eye(M.shape[1], dtype=dtype)).astype(dtype)
    return W, (y.T @ cb).astype(dtype), part


def pcg(W64, b64, L32, x0, tol=1e-6, max_it=12):
    """the kernel's loop: x += a p, r -= a W p, z = P^-1 r (two fp32 triangular solves), p = z + beta p"""
    def precond(r):
        z = np.linalg.solve(L32, r.astype(np.float32))
        return np.linalg.solve(L32.T, z.astype(np.float32)).astype(np.float64)
    x = x0.astype(np.float64)
    r = b64 - W64 @ x
    p = precond(r)
    rz = r @ p
    for _ in range(max_it):
        if not rz > 0:
            break
        wp = W64 @ p
        a = rz / (p @ wp)
        x += a * p
        r -= a * wp
        if not np.max(np.abs(a * p)) > tol * np.max(np.abs(x)):
            break
        z = precond(r)
        rz2 = r @ z
        p = z + (rz2 / rz) * p
        rz = rz2
    return x


def test_marked_rows_refined_by_preconditioned_cg_reach_the_reference():
    k, n_users, n_items, n_stale, r_csr, c_csr, Y0, cfg = draw_case(103)
    alpha, lam = cfg["alpha"], cfg["lam"]
    assert cfg["flags"] == 0
    Xo = oracle.half_iteration(*r_csr, Y0, alpha=alpha, lam=lam, threads=4)
    Yo = oracle.half_iteration(*c_csr, Xo, alpha=alpha, lam=lam, threads=4)
    G = Xo.astype(np.float64).T @ Xo.astype(np.float64)
    fast, refined, marked = np.zeros_like(Yo), np.zeros_like(Yo), 0
    for r in range(n_items):
        W32, b32, part32 = row_system(c_csr, Xo, G, r, alpha, lam, np.float32)
        L32 = np.linalg.cholesky(W32)                      # float32 throughout, like the kernels' tiles
        assert L32.dtype == np.float32
        x0 = np.linalg.solve(L32.T, np.linalg.solve(L32, b32)).astype(np.float32)
        fast[r] = x0
        est = max(float(part32.diagonal().max(initial=0.0)), 0.25 * float(W32.diagonal().max())) / float((L32.diagonal() ** 2).min())
        if est > 64.0:
            marked += 1
            W64, b64, _ = row_system(c_csr, Xo, G, r, alpha, lam, np.float64)
            refined[r] = pcg(W64, b64, L32, x0).astype(np.float32)
        else:
            refined[r] = x0
    rel = lambda a: float(np.linalg.norm(a.astype(np.float64) - Yo) / np.linalg.norm(Yo))   # noqa: E731
    assert marked > n_items // 2
    assert rel(fast) > 1e-4              # what fp32 alone does to these systems
    assert rel(refined) < 2e-6           # the marks + CG against the exact system


def test_fp64_restatement_with_the_reference_roundings_reaches_the_reference_where_exact_arithmetic_does_not():
    """als_exact_kernel's arithmetic on case 1085 (reconstructR + lossIgnoresUnspecified, lambda = 0.01, cond(W) ~ 1e7):
    W = sum over the row's entries of (double)(float)(y_r y_c) (ALS:524-539), b = sum r y, an fp64 LDL^T.  With the
    products rounded like the reference's the answer is the oracle's; with exact products it is 1e-2 away."""
    k, n_users, n_items, n_stale, r_csr, c_csr, Y0, cfg = draw_case(1085)
    assert cfg["flags"] == 3
    kw = dict(alpha=cfg["alpha"], lam=cfg["lam"], flags=3, threads=4)
    Xo = oracle.half_iteration(*r_csr, Y0, **kw)
    Yo = oracle.half_iteration(*c_csr, Xo, **kw)
    rp, col, val = c_csr
    rounded, exact = np.zeros_like(Yo), np.zeros_like(Yo)
    for r in range(n_items):
        a, b = rp[r], rp[r + 1]
        y32 = Xo[col[a:b]].astype(np.float32)
        rhs = y32.astype(np.float64).T @ val[a:b].astype(np.float64)                        # ALS:466-469
        ridge = cfg["lam"] * cfg["alpha"] * (b - a) * np.eye(k)
        W_ref = np.einsum("er,ec->erc", y32, y32).astype(np.float32).astype(np.float64).sum(0) + ridge   # float products
        W_exact = y32.astype(np.float64).T @ y32.astype(np.float64) + ridge
        rounded[r] = np.linalg.solve(W_ref, rhs)
        exact[r] = np.linalg.solve(W_exact, rhs)
    rel = lambda a: float(np.linalg.norm(a.astype(np.float64) - Yo) / np.linalg.norm(Yo))   # noqa: E731
    assert rel(rounded) < 1e-6
    assert rel(exact) > 1e-3
