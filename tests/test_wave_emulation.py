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
