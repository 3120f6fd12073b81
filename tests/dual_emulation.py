"""numpy restatement of the dual ("short row") solve path of csrc/dual_kernels.h, arithmetic step by step:
fp64 eigendecomposition of G, fp32 rotated rows, the d_f(n) table, the power-of-two operand scale, operands split
into two f16 halves (both rounded to nearest, as v_cvt_pk_f16_f32 does in the kernel), S = I + Z Z^T from the three
exact products in fp32, fp32 Cholesky + solves, x' = D^1/2 Z^T v, x = Q x'.  Test infrastructure (CPU): checked
against the oracle in tests/test_dual_emulation.py so that the ALGORITHM is validated without a GPU."""
import numpy as np


def split22(z):
    """z (fp32) -> (hi, lo): hi = z rounded to f16, lo = the exact residual z - hi rounded to f16 (pk_rn16 in the kernels;
    the operand scale keeps both inside f16's normal range)."""
    z = z.astype(np.float32)
    hi = z.astype(np.float16).astype(np.float32)
    lo = (z - hi).astype(np.float32).astype(np.float16).astype(np.float32)
    return hi, lo


def prepare(M, alpha, lam, eigh=np.linalg.eigh):
    """Once per half-iteration: G = M^T M (fp64), its eigendecomposition, the rotated copy, the z bound."""
    G = M.astype(np.float64).T @ M.astype(np.float64)
    L, Q = eigh(G)
    L = np.maximum(L, 0.0)
    Mr = (M.astype(np.float64) @ Q).astype(np.float32)
    dmax = (1.0 / np.sqrt(L + lam * alpha)).astype(np.float32)
    zbound = np.float32(np.max(np.abs(Mr) * dmax[None, :])) if len(M) else np.float32(0)
    return {"G": G, "L": L.astype(np.float32), "Q": Q, "Mr": Mr, "zbound": zbound}


def solve_row(prep, cols, vals, alpha, lam, w_max_sqrt):
    """One row with n <= k entries."""
    n = len(cols)
    k = prep["Mr"].shape[1]
    if n == 0:
        return np.zeros(k, dtype=np.float32)
    d = (np.float32(1.0) / np.sqrt(prep["L"] + np.float32(lam * alpha) * np.float32(n))).astype(np.float32)
    ar = (np.float32(alpha) * np.abs(vals.astype(np.float32))).astype(np.float32)
    w = np.sqrt(ar).astype(np.float32)
    cb = np.where(vals > 0, np.float32(1) + ar, np.float32(0)).astype(np.float32)
    q = np.where(w > 0, cb / np.where(w > 0, w, 1), 0).astype(np.float32)
    bound = np.float32(prep["zbound"]) * np.float32(w_max_sqrt)
    pw = 0 if not bound > 0 else int(np.clip(14 - int(np.floor(np.log2(bound)) + 1), -60, 60))
    sc = np.float32(2.0 ** pw)
    Y = prep["Mr"][cols]
    Z = ((Y * d[None, :]).astype(np.float32) * (w * sc)[:, None]).astype(np.float32)
    zh, zl = split22(Z)
    zh64, zl64 = zh.astype(np.float64), zl.astype(np.float64)
    S = (zh64 @ zh64.T + zh64 @ zl64.T + zl64 @ zh64.T).astype(np.float32)       # exact products, fp32 result
    S = (S * np.float32(2.0 ** (-2 * pw))).astype(np.float32) + np.eye(n, dtype=np.float32)
    Lc = np.linalg.cholesky(S.astype(np.float32)).astype(np.float32)
    v = np.linalg.solve(Lc.astype(np.float32), q).astype(np.float32)
    v = np.linalg.solve(Lc.T.astype(np.float32), v).astype(np.float32)
    xp = (((zh + zl).astype(np.float32).T @ v).astype(np.float32) * d * np.float32(2.0 ** (-pw))).astype(np.float32)
    return (prep["Q"] @ xp.astype(np.float64)).astype(np.float32)
