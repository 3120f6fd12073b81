"""Parity at BASELINE.json's full sizes through size-independent properties.

The oracle cannot run a whole C2/C4 half-iteration in test time, but every row of a half-iteration
is an independent function of (its entries, the opposite factors, their Gramian).  So at full size:
  * Gramian: linearity over row ranges (sum of partial Gramians = full Gramian, exactly the
    all-reduce the multi-GPU path relies on) + the oracle on a slice;
  * rows: a seeded random sample of rows AND the longest rows (segments path) recomputed by the
    oracle from the same inputs must equal the GPU rows (<= 1e-4 relative Frobenius);
  * every output finite (GenerationSerializer.java:195-197 asserts this of the reference).
"""
import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) /
                 max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def sub_csr(csr, rows, torch):
    rp = csr[0]
    rows_t = torch.as_tensor(rows, device=rp.device)
    lens = rp[rows_t + 1] - rp[rows_t]
    sub_rp = torch.zeros(len(rows) + 1, dtype=torch.int64, device=rp.device)
    torch.cumsum(lens, 0, out=sub_rp[1:])
    ent = torch.repeat_interleave(rp[rows_t] - sub_rp[:-1], lens) + torch.arange(int(sub_rp[-1]), device=rp.device)
    return sub_rp.cpu().numpy(), csr[1][ent].cpu().numpy(), csr[2][ent].cpu().numpy()


def check_half(core, side, csr, M_host, G, n_rows, rng, torch, n_sample=300, n_long=8):
    """Compare sampled + longest rows of `side` against the oracle."""
    lens = (csr[0][1:] - csr[0][:-1])
    longest = torch.topk(lens, n_long).indices.cpu().numpy()
    sample = rng.choice(n_rows, size=n_sample, replace=False)
    rows = np.unique(np.concatenate([sample, longest])).astype(np.int64)
    rp, col, val = sub_csr(csr, rows, torch)
    expect = oracle.solve_rows(rp, col, val, M_host, G, threads=8)
    got = core.get_rows(side, rows)
    assert np.all(np.isfinite(got))
    err = rel(got, expect)
    assert err < REL_TOL, (side, err)
    worst = max(rel(got[i], expect[i]) for i in range(len(rows)))
    assert worst < 20 * REL_TOL, (side, worst)
    return int(lens.max())


@pytest.mark.parametrize("name,n_users,n_items,nnz,k", [
    ("C1 MovieLens-100K shape", 943, 1_682, 100_000, 10),
    ("C2 MovieLens-25M shape", 162_541, 59_047, 25_000_095, 50),
    ("C3 Netflix-Prize shape", 480_189, 17_770, 100_480_507, 100),
    ("C4 synthetic 10M x 1M", 10_000_000, 1_000_000, 1_000_000_000, 64),
    # C5 (100M x 10M, 5e9 entries, k=128) is an 8-GPU configuration: one rank's share of the user rows
    # against the full item side -- the shapes, row lengths and kernels (k = 128: dual path for the short
    # rows, T = 8 direct kernels for the long ones) of a C5 rank
    ("C5 one-rank shard (1/8 of the users)", 12_500_000, 10_000_000, 625_000_000, 128),
])
def test_full_size_half_iterations(name, n_users, n_items, nnz, k):
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1234567890)
    prob = synth.torch_problem(n_users, n_items, nnz, k, dev)
    with pkg.ALSCore(k, device=0) as core:
        core.set_factor_rows(pkg.SIDE_X, n_users)
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_matrix(pkg.SIDE_X, *prob["r_csr"])
        core.set_matrix(pkg.SIDE_Y, *prob["c_csr"])
        Y0 = prob["Y0"].cpu().numpy()
        core.set_factors(pkg.SIDE_Y, Y0)

        # --- Gramian of Y0: GPU fp64 vs oracle on the whole matrix (up to 1M rows; a 1M-row slice beyond)
        Gy = core.gramian(pkg.SIDE_Y, fetch=True)
        if n_items <= 1_000_000:
            assert rel(Gy, oracle.gramian(Y0)) < 5e-7
        else:
            gs = torch.zeros(k, k, dtype=torch.float64, device=dev)
            core.gramian_partial(pkg.SIDE_Y, 12_345, 1_000_000, gs)
            torch.cuda.synchronize()
            assert rel(gs.cpu().numpy(), oracle.gramian(Y0[12_345:1_012_345])) < 5e-7
        core.reset_stats()
        # --- X half
        core.solve_side(pkg.SIDE_X)
        core.check()
        max_len_x = check_half(core, pkg.SIDE_X, prob["r_csr"], Y0, Gy, n_users, rng, torch)

        # --- Gramian of X: linearity over row ranges (what the k x k all-reduce relies on) + oracle on a slice
        Gx = core.gramian(pkg.SIDE_X, fetch=True)
        parts = torch.zeros(3, k, k, dtype=torch.float64, device=dev)
        cuts = [0, n_users // 3, 2 * n_users // 3 + 5, n_users]
        for i in range(3):
            core.gramian_partial(pkg.SIDE_X, cuts[i], cuts[i + 1] - cuts[i], parts[i])
        torch.cuda.synchronize()
        # below 262144 rows every piece runs the fp64 kernel (exact products, fp64 sums): 1e-12; above, the whole
        # matrix runs the split-f16 kernel (fp32 sums inside 2048-row slabs, fp64 across): <= 2e-7
        assert rel(parts.sum(0).cpu().numpy(), Gx) < (1e-12 if n_users < 262144 else 2e-7)
        n_slice = min(n_users, 200_000)
        Xs = core.get_factors(pkg.SIDE_X, 0, n_slice)
        gs = torch.zeros(k, k, dtype=torch.float64, device=dev)
        core.gramian_partial(pkg.SIDE_X, 0, n_slice, gs)
        torch.cuda.synchronize()
        assert rel(gs.cpu().numpy(), oracle.gramian(Xs)) < 5e-7
        assert np.allclose(Gx, Gx.T)

        # --- Y half (long rows: exercises the segments + finish path at C4)
        X = core.get_factors(pkg.SIDE_X)
        assert np.all(np.isfinite(X))
        core.solve_side(pkg.SIDE_Y)
        core.check()
        max_len_y = check_half(core, pkg.SIDE_Y, prob["c_csr"], X, Gx, n_items, rng, torch)
        if "C4" in name or "C3" in name:
            assert max_len_y > 4096, "C3 / C4 must exercise the long-row (segments) path"
        st = core.stats()
        assert st["rows_solved"] == n_users + n_items
        if k > 32:
            assert st["rows_dual"] > 0, "k > 32: the short rows go through the dual kernels"
        Y = core.get_factors(pkg.SIDE_Y)
        assert np.all(np.isfinite(Y))
        # a solved factor matrix is not degenerate
        assert np.linalg.norm(Y) > 0 and np.linalg.norm(X) > 0
        del max_len_x
