"""Validates the wave-level algorithm of the HIP kernel (fragment layouts, blocked Cholesky,
triangular solves) against the oracle, using the lane-level numpy emulation."""
import numpy as np
import pytest

from oracle import oracle
from tests import wave_emulation as we


@pytest.mark.parametrize("k,n_u", [(2, 3), (10, 17), (16, 5), (30, 40), (50, 33), (64, 20)])
def test_wave_algorithm_matches_oracle(k, n_u):
    rng = np.random.default_rng(100 * k + n_u)
    n_m = 300
    M = (rng.standard_normal((n_m, k)) / np.sqrt(k)).astype(np.float32)
    G = oracle.gramian(M)
    cols = rng.choice(n_m, size=n_u, replace=False).astype(np.int32)
    vals = rng.choice([-2.0, 1.0, 2.0, 3.5, 5.0], size=n_u).astype(np.float32)
    row_ptr = np.array([0, n_u], dtype=np.int64)
    for flags, kw in [(0, {}), (oracle.FLAG_RECONSTRUCT_R, {"reconstruct": True})]:
        ref = oracle.solve_rows(row_ptr, cols, vals, M, G, flags=flags)[0]
        x, minpiv = we.wave_solve_row(M[cols], vals, G, k, **kw)
        assert minpiv > 0
        err = np.linalg.norm(x - ref) / max(np.linalg.norm(ref), 1e-30)
        assert err < 2e-5, (k, n_u, flags, err)


def test_wave_algorithm_k100_blocks():
    k, n_u = 100, 12
    rng = np.random.default_rng(5)
    M = (rng.standard_normal((200, k)) / np.sqrt(k)).astype(np.float32)
    G = oracle.gramian(M)
    cols = np.arange(n_u, dtype=np.int32)
    vals = np.ones(n_u, np.float32)
    ref = oracle.solve_rows(np.array([0, n_u], np.int64), cols, vals, M, G)[0]
    x, _ = we.wave_solve_row(M[cols], vals, G, k)
    assert np.linalg.norm(x - ref) / np.linalg.norm(ref) < 2e-5


@pytest.mark.parametrize("k,n_u,alpha,vscale", [(64, 100, 1.0, 1.0), (50, 33, 40.0, 1.0), (33, 70, 1.0, 1000.0),
                                              (48, 5, 1.0, 1e-3), (64, 32, 1.0, 1.0)])
def test_split_f16_gather_matches_oracle(k, n_u, alpha, vscale):
    """MALS_GRAMIAN_SPLIT_F16 (gather_row_h): operands split into two f16 halves, exact products."""
    rng = np.random.default_rng(7 * k + n_u)
    n_m = 400
    M = (rng.standard_normal((n_m, k)) / np.sqrt(k)).astype(np.float32)
    G = oracle.gramian(M)
    cols = rng.choice(n_m, size=n_u, replace=False).astype(np.int32)
    vals = (rng.choice([-2.0, 1.0, 2.0, 3.5, 5.0], size=n_u) * vscale).astype(np.float32)
    row_ptr = np.array([0, n_u], dtype=np.int64)
    for flags, kw in [(0, {}), (oracle.FLAG_RECONSTRUCT_R, {"reconstruct": True}),
                      (oracle.FLAG_LOSS_IGNORES_UNSPECIFIED, {"loss_ignores": True})]:
        if flags == oracle.FLAG_LOSS_IGNORES_UNSPECIFIED and n_u < k:
            continue                                    # W = sum over n_u < k outer products: singular
        ref = oracle.solve_rows(row_ptr, cols, vals, M, G, alpha=alpha, flags=flags)[0]
        x, minpiv = we.wave_solve_row(M[cols], vals, G, k, alpha=alpha, split_f16=True, max_abs_val=5.0 * vscale, **kw)
        assert minpiv > 0
        err = np.linalg.norm(x - ref) / max(np.linalg.norm(ref), 1e-30)
        assert err < 2e-5, (k, n_u, flags, err)


def test_split_f16_operands_never_overflow_and_keep_22_bits():
    rng = np.random.default_rng(3)
    y = (rng.standard_normal(4096) * 3).astype(np.float32)
    G = np.array([[float(np.sum(y.astype(np.float64) ** 2))]])
    w_max = 1.0 + 40.0 * 1000.0
    S, inv_s2 = we.gather_scale(G, 1, w_max)
    z = (y * np.float32(np.sqrt(w_max)) * S).astype(np.float32)
    assert np.max(np.abs(z)) <= 2.0 ** 14 and float(S) * float(S) * float(inv_s2) == 1.0
    h = (z.view(np.uint32) & np.uint32(0xffffe000)).view(np.float32)
    zh, zl = we.f16_rn(h), we.f16_rn(z - h)
    assert np.all(zh == h)                                # the masked top bits are exact in f16
    big = np.abs(z) > 2.0 ** -3                           # full precision down to 2^-17 of the largest
    assert np.all(np.abs(z - (zh + zl))[big] <= np.abs(z[big]) * 2.0 ** -21)
