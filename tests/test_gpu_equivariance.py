"""A size-independent property of `Worker.call` (ALS:432-504): a row's solution depends on its own entries, the opposite factors
and their Gramian -- not on where the row sits in the matrix.  Solving the same half-iteration with the rows in another order must
therefore give the same factors, permuted, BIT FOR BIT: which wave solves a row, what it solved before it, whose first super-step
it prefetched, which chunk or segment list the row landed in are all supposed to be invisible.  Run at sizes where every path is
live (direct, dual, segments + finish, LDS-staged at k = 128), from C2 / C3-sized problems up to C4 at its full size: a
cross-row leak anywhere in the prefetch chains shows up as a single differing row."""
import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import synth

pytestmark = pytest.mark.gpu


def permute_rows(torch, row_ptr, col, val, perm):
    """CSR with row i = old row perm[i] (entries of a row keep their order)."""
    lens = (row_ptr[1:] - row_ptr[:-1])[perm]
    new_ptr = torch.zeros_like(row_ptr)
    torch.cumsum(lens, 0, out=new_ptr[1:])
    nnz = int(new_ptr[-1])
    src = torch.repeat_interleave(row_ptr[:-1][perm] - new_ptr[:-1], lens) + torch.arange(nnz, device=row_ptr.device, dtype=torch.int64)
    return new_ptr, col[src], val[src]


def solve_side(torch, k, side, n_rows, n_opp, csr, opp_factors, **kw):
    with pkg.ALSCore(k, device=0, **kw) as core:
        core.set_factor_rows(side, n_rows)
        core.set_factor_rows(1 - side, n_opp)
        core.set_matrix(side, *csr)
        core.set_factors(1 - side, opp_factors)
        core.reset_stats()
        core.half_iteration(side)
        core.check()
        return core.get_factors(side), core.stats()


# (the first case is C4 at BASELINE.json's full size: 10M x 1M, 1e9 entries)
@pytest.mark.parametrize("k,n_users,n_items,nnz,chunk_rows", [(64, 10_000_000, 1_000_000, 1_000_000_000, 0), (64, 400_000, 60_000, 20_000_000, 70_001),
                                                              (128, 2_000_000, 200_000, 200_000_000, 0), (50, 200_000, 40_000, 10_000_000, 0),
                                                              (100, 480_000, 17_770, 100_000_000, 0)])
def test_row_order_is_invisible(k, n_users, n_items, nnz, chunk_rows):
    import torch
    dev = torch.device("cuda", 0)
    prob = synth.torch_problem(n_users, n_items, nnz, k, dev)
    Y0 = prob["Y0"].cpu().numpy()
    kw = dict(chunk_rows=chunk_rows) if chunk_rows else {}
    g = torch.Generator(device=dev)
    g.manual_seed(2024)
    # user half: short rows (dual), ordinary rows (direct / LDS-staged)
    X, st = solve_side(torch, k, pkg.SIDE_X, n_users, n_items, prob["r_csr"], Y0, **kw)
    perm = torch.randperm(n_users, generator=g, device=dev)
    pr = permute_rows(torch, *prob["r_csr"], perm)
    Xp, stp = solve_side(torch, k, pkg.SIDE_X, n_users, n_items, pr, Y0, **kw)
    del pr
    torch.cuda.empty_cache()
    assert st["rows_dual"] > 0 and st["rows_dual"] == stp["rows_dual"]
    assert np.all(np.isfinite(X)) and float(np.abs(X).max()) > 0.0 and not np.array_equal(Xp, X)      # (the comparison below is not vacuous)
    assert np.array_equal(Xp, X[perm.cpu().numpy()]), int(np.argmax(np.any(Xp != X[perm.cpu().numpy()], axis=1)))
    # item half: the popular items through segments + finish; gathers from X
    Y, _ = solve_side(torch, k, pkg.SIDE_Y, n_items, n_users, prob["c_csr"], X, **kw)
    permi = torch.randperm(n_items, generator=g, device=dev)
    pc = permute_rows(torch, *prob["c_csr"], permi)
    Yp, _ = solve_side(torch, k, pkg.SIDE_Y, n_items, n_users, pc, X, **kw)
    del pc
    torch.cuda.empty_cache()
    lens = (prob["c_csr"][0][1:] - prob["c_csr"][0][:-1])
    assert int(lens.max()) > 4096, "some item rows take the long-row path"
    assert np.array_equal(Yp, Y[permi.cpu().numpy()]), int(np.argmax(np.any(Yp != Y[permi.cpu().numpy()], axis=1)))


@pytest.mark.parametrize("k,n_users,n_items,nnz", [(64, 10_000_000, 1_000_000, 1_000_000_000), (128, 2_000_000, 200_000, 200_000_000)])
def test_a_full_iteration_run_twice_is_bit_identical(k, n_users, n_items, nnz):
    """Idempotence at full size: fixed work assignment, fixed summation orders (the Gramian's slab partials are summed in slab
    order), no float atomics -- two handles fed the same problem produce the same bits after a whole iteration, Gramians included."""
    import torch
    dev = torch.device("cuda", 0)
    prob = synth.torch_problem(n_users, n_items, nnz, k, dev)
    out = []
    for _ in range(2):
        with pkg.ALSCore(k, device=0) as core:
            core.set_factor_rows(pkg.SIDE_X, n_users)
            core.set_factor_rows(pkg.SIDE_Y, n_items)
            core.set_matrix(pkg.SIDE_X, *prob["r_csr"])
            core.set_matrix(pkg.SIDE_Y, *prob["c_csr"])
            core.set_factors(pkg.SIDE_Y, prob["Y0"].cpu().numpy())
            core.half_iteration(pkg.SIDE_X)
            core.half_iteration(pkg.SIDE_Y)
            core.check()
            G = core.gramian(pkg.SIDE_Y, fetch=True)
            out.append((core.get_factors(pkg.SIDE_X), core.get_factors(pkg.SIDE_Y), G))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert np.all(np.isfinite(out[0][1])) and float(np.abs(out[0][1]).max()) > 0.0 and out[0][2].shape == (k, k) and out[0][2][0, 0] > 0.0
