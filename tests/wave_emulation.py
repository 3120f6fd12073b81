"""Lane-level numpy emulation of the K2/K3 wave algorithm (one 64-lane wave per row).

This is a design check, not product code and not the oracle: it executes the exact register/lane
choreography the HIP kernel uses (myrrix-recommender_amd/csrc/als_kernels.h) with numpy arrays
of shape [64] standing in for VGPRs, so the MFMA fragment layouts, the blocked Cholesky and the
triangular solves can be validated on a CPU-only box (tests/test_wave_emulation.py).

Layouts (gfx950, cdna_hip_programming.md section 3):
  v_mfma_f32_16x16x4_f32: lane l supplies A[i=l&15][kk=l>>4] and B[kk=l>>4][j=l&15];
  C/D "acc layout": lane l, reg r holds D[row=4*(l>>4)+r][col=l&15].
"""
import numpy as np

L = np.arange(64)
G_ = L >> 4   # lane group 0..3
C_ = L & 15   # column within tile


def mfma_16x16x4(a, b, acc):
    """acc: [4][64] float32 in acc layout; a, b: [64].  Returns new acc (fp32 fma chain)."""
    A = a.reshape(4, 16).T          # A[i][kk] = a[16*kk + i]
    B = b.reshape(4, 16)            # B[kk][j] = b[16*kk + j]
    D = np.zeros((16, 16), dtype=np.float32)
    for row in range(16):
        D[row] = acc[row & 3][16 * (row >> 2):16 * (row >> 2) + 16]
    for kk in range(4):             # k-ordered fmaf chain (one rounding per product+add)
        D = (D.astype(np.float64) + np.outer(A[:, kk], B[kk]).astype(np.float64)).astype(np.float32)
    out = np.zeros((4, 64), dtype=np.float32)
    for row in range(16):
        out[row & 3][16 * (row >> 2):16 * (row >> 2) + 16] = D[row]
    return out


def shfl(x, idx):
    return x[idx]


def reduce_groups(x):
    """sum over the 4 lane groups (same column): x += shfl_xor(x,16); x += shfl_xor(x,32)."""
    x = x + x[L ^ 16]
    x = x + x[L ^ 32]
    return x


def reduce_row16(x):
    """sum over the 16 lanes of each lane row via DPP row_ror 8,4,2,1."""
    for n in (8, 4, 2, 1):
        x = x + x[(L & 48) | ((C_ + n) & 15)]
    return x


def tile_to_dense(t):
    D = np.zeros((16, 16), dtype=np.float32)
    for row in range(16):
        D[row] = t[row & 3][16 * (row >> 2):16 * (row >> 2) + 16]
    return D


def dense_to_tile(D):
    out = np.zeros((4, 64), dtype=np.float32)
    for row in range(16):
        out[row & 3][16 * (row >> 2):16 * (row >> 2) + 16] = D[row]
    return out


def wave_gram_rhs(Yg, w, cb, k):
    """Gather phase.  Yg: [n_u][k] gathered opposing rows (fp32), w: [n_u] Gramian weights,
    cb: [n_u] RHS weights.  Returns (acc[ti][tj] tiles for ti<=tj, bcol[v])."""
    T = (k + 15) // 16
    n_u = Yg.shape[0]
    acc = {(i, j): np.zeros((4, 64), np.float32) for i in range(T) for j in range(i, T)}
    bpart = [np.zeros(64, np.float32) for _ in range(T)]
    steps = (n_u + 3) // 4
    for s in range(steps):
        n = 4 * s + G_
        valid_n = n < n_u
        nn = np.where(valid_n, n, 0)
        wn = np.where(valid_n, w[nn] if n_u else 0.0, 0.0).astype(np.float32)
        cbn = np.where(valid_n, cb[nn] if n_u else 0.0, 0.0).astype(np.float32)
        yv = []
        for v in range(T):
            f = 16 * v + C_
            ok = valid_n & (f < k)
            yv.append(np.where(ok, Yg[nn, np.minimum(f, k - 1)] if n_u else 0.0, 0.0).astype(np.float32))
        a = [(wn * yv[v]).astype(np.float32) for v in range(T)]
        for i in range(T):
            for j in range(i, T):
                acc[(i, j)] = mfma_16x16x4(a[i], yv[j], acc[(i, j)])
        for v in range(T):
            bpart[v] = (bpart[v] + cbn * yv[v]).astype(np.float32)
    bcol = [reduce_groups(bpart[v]) for v in range(T)]
    return acc, bcol


def f16_rn(x):
    """v_cvt_pk_f16_f32 on each element: round to nearest even to f16 (returned as fp32 values).  (Until round 4 the kernels
    truncated -- v_cvt_pkrtz_f16_f32 --, which biased every operand towards zero: csrc/als_kernels.h, pk_rn16.)"""
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def gather_scale(Gd, k, w_max):
    """gather_scale_kernel: S = 2^e with sqrt(w_max) * S * sqrt(max_f G_ff) < 2^14."""
    bound = np.sqrt(max(float(np.max(np.diag(Gd)[:k])), 0.0)) * np.sqrt(float(w_max))
    e = 0
    if 0.0 < bound < 1e300:
        e = int(np.clip(14 - np.frexp(bound)[1], -60, 60))
    return np.float32(np.ldexp(1.0, e)), np.float32(np.ldexp(1.0, -2 * e))


def mfma_16x16x32_f16(a, b, acc):
    """acc: [4][64] fp32 acc layout; a, b: [8][64] f16-representable values, slot e of lane (g,c) =
    contraction index 8g+e.  Products of f16 numbers are exact in fp32; the hardware's internal
    summation order is not documented, so the 32-term dot is taken exactly (fp64) and rounded once."""
    A = np.zeros((16, 32), np.float64)
    B = np.zeros((32, 16), np.float64)
    for e in range(8):
        for g in range(4):
            A[:, 8 * g + e] = a[e][16 * g:16 * g + 16]
            B[8 * g + e, :] = b[e][16 * g:16 * g + 16]
    D = A @ B
    out = np.zeros((4, 64), dtype=np.float32)
    for row in range(16):
        sl = slice(16 * (row >> 2), 16 * (row >> 2) + 16)
        out[row & 3][sl] = (acc[row & 3][sl].astype(np.float64) + D[row]).astype(np.float32)
    return out


def wave_gram_rhs_split(Yg, w, cb, k, zscale, inv_s2):
    """Split-precision gather (gather_row_h): super-steps of 32 entries, lane (g,c) slot e = entry
    4e+g; z = sqrt(w)*S*y split as zh (rounded to f16) + zl (the rest, rounded to f16);
    acc += zh zh^T + zh zl^T + zl zh^T on the f16 matrix pipe; RHS from the raw rows in fp32.
    Returns the UNSCALED tiles (acc / S^2) and bcol."""
    T = (k + 15) // 16
    n_u = Yg.shape[0]
    acc = {(i, j): np.zeros((4, 64), np.float32) for i in range(T) for j in range(i, T)}
    bpart = [np.zeros(64, np.float32) for _ in range(T)]
    sw = (np.sqrt(np.asarray(w, np.float32)) * np.float32(zscale)).astype(np.float32)
    for ss in range((n_u + 31) // 32):
        zh = [[None] * 8 for _ in range(T)]
        zl = [[None] * 8 for _ in range(T)]
        for e in range(8):
            n = 32 * ss + 4 * e + G_
            valid_n = n < n_u
            nn = np.where(valid_n, n, n_u - 1)                    # clamped column, zero weights
            swn = np.where(valid_n, sw[nn], 0.0).astype(np.float32)
            cbn = np.where(valid_n, cb[nn], 0.0).astype(np.float32)
            for v in range(T):
                f = 16 * v + C_
                y = np.where(f < k, Yg[nn, np.minimum(f, k - 1)], 0.0).astype(np.float32)
                z = (y * swn).astype(np.float32)
                zh[v][e] = f16_rn(z)                              # v_cvt_pk_f16_f32
                # v_fma_mix_f32: the residual of the EXACT product y * sqrt(w) S, rounded once
                zl[v][e] = f16_rn((y.astype(np.float64) * swn.astype(np.float64) - zh[v][e]).astype(np.float32))
                bpart[v] = (bpart[v] + cbn * y).astype(np.float32)
        for a_, b_ in ((zh, zh), (zh, zl), (zl, zh)):
            for i in range(T):
                for j in range(i, T):
                    acc[(i, j)] = mfma_16x16x32_f16(a_[i], b_[j], acc[(i, j)])
    for key in acc:
        acc[key] = (acc[key] * np.float32(inv_s2)).astype(np.float32)
    bcol = [reduce_groups(bpart[v]) for v in range(T)]
    return acc, bcol


def wave_add_base(acc, Gd, ridge, k, use_g=True):
    """acc += G (acc-layout image of the fp32 Gramian), += ridge on the diagonal, 1 on padding."""
    T = (k + 15) // 16
    Gp = np.zeros((16 * T, 16 * T), np.float32)
    if use_g:
        Gp[:k, :k] = Gd.astype(np.float32)
    for i in range(T):
        for j in range(i, T):
            acc[(i, j)] = (acc[(i, j)] + dense_to_tile(Gp[16 * i:16 * i + 16, 16 * j:16 * j + 16])).astype(np.float32)
    for v in range(T):
        t = acc[(v, v)]
        for r in range(4):
            diag = (4 * G_ + r) == C_
            feat = 16 * v + C_
            t[r] = np.where(diag & (feat < k), t[r] + np.float32(ridge), t[r])
            t[r] = np.where(diag & (feat >= k), np.float32(1.0), t[r])
    return acc


def row_bcast(x, lane_in_row):
    """DPP row_newbcast: every lane of a 16-lane row reads lane `lane_in_row` of its own row."""
    return x[(L & 48) | lane_in_row]


def wave_factor_diag(D):
    """In-tile factorization of a full symmetric 16x16 tile D (acc layout) by LDL^T-style elimination
    on [D | I]: lane (g,c) owns row c, columns 4g..4g+3 of both halves (D read through its symmetry).
    Returns (Uinv, minpiv): Uinv = U^{-1} in acc layout, where U^T U = D."""
    D = D.copy()
    E = np.zeros((4, 64), np.float32)
    for r in range(4):
        E[r] = np.where((4 * G_ + r) == C_, 1.0, 0.0)
    minpiv = np.inf
    mypiv = np.ones(64, np.float32)
    pos = 4 * (C_ & 3) + (C_ >> 2)                 # the step at which row c is the pivot row
    for t in range(16):                            # pivot order: register by register (see diag_step)
        gm, rm = t & 3, t >> 2
        m = 4 * gm + rm
        piv = D[rm][16 * gm + m]                   # v_readlane (static lane)
        minpiv = min(minpiv, piv)
        rinv = np.float32(1.0) / np.float32(piv)
        num = shfl(D[rm], 16 * gm + C_)            # D[c][m]: the one cross-group move of the step
        nl = np.where(pos > t, -(num * rinv).astype(np.float32), np.float32(0))
        mypiv = np.where(pos == t, np.float32(piv), mypiv)
        for r in range(4):
            if 4 * r + 3 > t:                      # register r of the D half is finished after step 4r+3
                D[r] = (D[r] + row_bcast(D[r], m) * nl).astype(np.float32)
            if 4 * r <= t:                         # register r of the inverse half is zero before step 4r
                E[r] = (E[r] + row_bcast(E[r], m) * nl).astype(np.float32)
    s = (np.float32(1.0) / np.sqrt(mypiv)).astype(np.float32)
    return (E * s).astype(np.float32), minpiv


def mfma_16x16x16_f16(a, b, acc):
    """acc: [4][64] fp32 acc layout; a, b: [4][64] f16-representable values, reg r of lane (g,c) =
    contraction index 4g+r (an acc-layout tile used as operand: D = P^T Q).  Exact products, the
    16-term dot taken exactly and rounded once (see mfma_16x16x32_f16)."""
    P = np.zeros((16, 16), np.float64)   # P[k][i]
    Q = np.zeros((16, 16), np.float64)   # Q[k][j]
    for r in range(4):
        for g in range(4):
            P[4 * g + r, :] = a[r][16 * g:16 * g + 16]
            Q[4 * g + r, :] = b[r][16 * g:16 * g + 16]
    D = P.T @ Q
    out = np.zeros((4, 64), dtype=np.float32)
    for row in range(16):
        sl = slice(16 * (row >> 2), 16 * (row >> 2) + 16)
        out[row & 3][sl] = (acc[row & 3][sl].astype(np.float64) + D[row]).astype(np.float32)
    return out


def split_tile(t):
    """split_tile: every register of an acc-layout tile -> (top 11 significand bits, next 11 toward zero)."""
    t = np.asarray(t, np.float32)
    h = f16_rn(t)
    return h, f16_rn((t - h).astype(np.float32))


def row_scale(acc, bcol, T):
    """row_scale: s^2 = 2^(2p) with s * sqrt(max diag) <= 2^13; returns 1/s^2.  The largest diagonal
    element is taken as the largest |entry| of the diagonal tiles (equal for an SPD matrix)."""
    m = np.float32(0)
    for v in range(T):
        m = max(m, np.float32(np.max(np.abs(acc[(v, v)]))))
    e = ((int(np.float32(m).view(np.uint32)) >> 23) & 255) - 126
    p2 = int(np.clip(2 * (13 - ((e + 1) >> 1)), -100, 100))
    s2, inv_s2 = np.float32(np.ldexp(1.0, p2)), np.float32(np.ldexp(1.0, -p2))
    for key in acc:
        acc[key] = (acc[key] * s2).astype(np.float32)
    return [(b * s2).astype(np.float32) for b in bcol], inv_s2


def wave_cholesky(acc, T, split_syrk=False):
    """Blocked right-looking Cholesky on the upper tiles.  On return acc[(i,j)], i<j hold U tiles
    and acc[(i,i)] hold Uinv_ii.  split_syrk: cholesky_tiles<T, true> (the rank-16 updates with
    split f16 operands; the caller has applied row_scale)."""
    minpiv = np.inf
    zero = np.zeros((4, 64), np.float32)
    for kb in range(T):
        Uinv, mp = wave_factor_diag(acc[(kb, kb)])
        minpiv = min(minpiv, mp)
        acc[(kb, kb)] = Uinv
        for j in range(kb + 1, T):                 # TRSM: U_kj = Uinv^T A_kj
            Q = acc[(kb, j)]
            new = zero.copy()
            for r in range(4):
                new = mfma_16x16x4(Uinv[r], Q[r], new)
            acc[(kb, j)] = new
        q = {j: split_tile(acc[(kb, j)]) for j in range(kb + 1, T)} if split_syrk else None
        for i in range(kb + 1, T):                 # SYRK: A_ij -= U_ki^T U_kj
            for j in range(i, T):
                P, Q = acc[(kb, i)], acc[(kb, j)]
                t = acc[(i, j)]
                if split_syrk:
                    nh, nl = -q[i][0], -q[i][1]
                    t = mfma_16x16x16_f16(nh, q[j][0], t)
                    t = mfma_16x16x16_f16(nh, q[j][1], t)
                    t = mfma_16x16x16_f16(nl, q[j][0], t)
                else:
                    for r in range(4):
                        t = mfma_16x16x4(-P[r], Q[r], t)
                acc[(i, j)] = t
    return acc, minpiv


def col_to_row(vcol):
    """col layout (lane c holds v[c]) -> row layout regs r: v[4g+r]."""
    return [shfl(vcol, (L & 48) | (4 * G_ + r)) for r in range(4)]


def row_to_col(vrow):
    """row layout (all lanes of group g hold v[4g+r] in reg r) -> col layout."""
    t = [shfl(vrow[q], 16 * (C_ >> 2)) for q in range(4)]
    return np.select([(C_ & 3) == q for q in range(4)], t)


def quad_perm(x, perm):
    """DPP quad_perm: lane 4q+i reads lane 4q+perm[i]."""
    return x[(L & ~3) | np.asarray(perm)[L & 3]]


def row_ror(x, n):
    """DPP row_ror: lane c of a 16-lane row reads lane (c + n) % 16 of the row."""
    return x[(L & 48) | ((L + n) & 15)]


def reduce4_row16(v):
    """Sum each of four registers over the 16 lanes of a row; lane c ends with the total of register c & 3."""
    b0, b1 = (L & 1) != 0, (L & 2) != 0
    a0 = (np.where(b0, v[1], v[0]) + quad_perm(np.where(b0, v[0], v[1]), [1, 0, 3, 2])).astype(np.float32)
    a1 = (np.where(b0, v[3], v[2]) + quad_perm(np.where(b0, v[2], v[3]), [1, 0, 3, 2])).astype(np.float32)
    t = (np.where(b1, a1, a0) + quad_perm(np.where(b1, a0, a1), [2, 3, 0, 1])).astype(np.float32)
    t = (t + row_ror(t, 4)).astype(np.float32)
    t = (t + row_ror(t, 8)).astype(np.float32)
    return t


def spread_to_col(v):
    """lane (g,c) holds element 4g + (c&3) of a row-indexed vector -> col layout (lane (g,c): element c)."""
    return shfl(v, 16 * (C_ >> 2) + C_)


def wave_solve(acc, bcol, T):
    """x = W^{-1} b with W = U^T U; acc as returned by wave_cholesky.  Returns xcol[v]."""
    zrow, zcol = [], []
    for kb in range(T):                            # forward: z = U^{-T} b
        t = np.zeros(64, np.float32)
        for i in range(kb):
            for r in range(4):
                t = (t + acc[(i, kb)][r] * zrow[i][r]).astype(np.float32)
        rhs = (bcol[kb] - reduce_groups(t)).astype(np.float32)
        rr = col_to_row(rhs)
        zt = np.zeros(64, np.float32)
        for r in range(4):
            zt = (zt + acc[(kb, kb)][r] * rr[r]).astype(np.float32)
        zcol.append(reduce_groups(zt))
        zrow.append(col_to_row(zcol[kb]))
    xcol = [None] * T
    for kb in range(T - 1, -1, -1):                # backward: x = U^{-1} z
        rhs_col = zcol[kb]
        if kb < T - 1:
            t = [np.zeros(64, np.float32) for _ in range(4)]
            for r in range(4):
                for j in range(kb + 1, T):
                    t[r] = (t[r] + acc[(kb, j)][r] * xcol[j]).astype(np.float32)
            rhs_col = (rhs_col - spread_to_col(reduce4_row16(t))).astype(np.float32)
        xr = [(acc[(kb, kb)][r] * rhs_col).astype(np.float32) for r in range(4)]
        xcol[kb] = spread_to_col(reduce4_row16(xr))
    return xcol


def wave_solve_row(Yg, vals, Gd, k, alpha=1.0, lam=0.1, reconstruct=False, loss_ignores=False, split_f16=False,
                   max_abs_val=None):
    """Full per-row pipeline; returns x[k] (fp32) and the smallest Cholesky pivot.  split_f16: the
    MALS_GRAMIAN_SPLIT_F16 gather (max_abs_val = largest |value| of the whole matrix side)."""
    vals = np.asarray(vals, np.float32)
    n_u = len(vals)
    base_w = 1.0 if loss_ignores else 0.0
    if reconstruct:
        w = np.full(n_u, base_w, np.float32)
        cb = vals.copy()
    else:
        w = (base_w + alpha * np.abs(vals)).astype(np.float32)
        cb = np.where(vals > 0, 1.0 + alpha * np.abs(vals), 0.0).astype(np.float32)
    T = (k + 15) // 16
    if split_f16 and n_u:
        mx = float(np.max(np.abs(vals))) if max_abs_val is None else float(max_abs_val)
        zscale, inv_s2 = gather_scale(Gd, k, base_w + (0.0 if reconstruct else abs(alpha) * mx))
        acc, bcol = wave_gram_rhs_split(np.asarray(Yg, np.float32).reshape(n_u, k), w, cb, k, zscale, inv_s2)
    else:
        acc, bcol = wave_gram_rhs(np.asarray(Yg, np.float32).reshape(n_u, k), w, cb, k)
    acc = wave_add_base(acc, Gd, lam * alpha * n_u, k, use_g=not loss_ignores)
    if split_f16 and T >= 2:                       # the split-precision kernel scales the row and splits the SYRK too
        bcol, inv_s2 = row_scale(acc, bcol, T)
        acc, minpiv = wave_cholesky(acc, T, split_syrk=True)
        minpiv = minpiv * inv_s2
    else:
        acc, minpiv = wave_cholesky(acc, T)
    xcol = wave_solve(acc, bcol, T)
    x = np.zeros(16 * T, np.float32)
    for v in range(T):
        x[16 * v:16 * v + 16] = xcol[v][:16]
    return x[:k], minpiv
