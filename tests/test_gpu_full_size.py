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


def independent_gramian(M, torch, dev, chunk=1_000_000):
    """M^T M in fp64 by torch (a matmul of the widened rows, chunk by chunk) -- a checker that shares no code with the
    library's Gramian kernels: what the oracle is handed where its own O(n k^2) pass does not finish in test time."""
    G = None
    n = M.shape[0]
    for r0 in range(0, n, chunk):
        blk = M[r0:r0 + chunk]
        blk = (blk if torch.is_tensor(blk) else torch.as_tensor(blk)).to(dev).double()
        part = blk.T @ blk
        G = part if G is None else G + part
    return G.cpu().numpy()


def sub_csr(csr, rows, torch):
    rp = csr[0]
    rows_t = torch.as_tensor(rows, device=rp.device)
    lens = rp[rows_t + 1] - rp[rows_t]
    sub_rp = torch.zeros(len(rows) + 1, dtype=torch.int64, device=rp.device)
    torch.cumsum(lens, 0, out=sub_rp[1:])
    ent = torch.repeat_interleave(rp[rows_t] - sub_rp[:-1], lens) + torch.arange(int(sub_rp[-1]), device=rp.device)
    return sub_rp.cpu().numpy(), csr[1][ent].cpu().numpy(), csr[2][ent].cpu().numpy()


# Row-length classes = the kernel paths a row can take: empty rows, the dual classes (1-16 ... 49-64
# entries at k > 32), direct rows of a few super-steps, of many, rows up to the segment length, rows cut into segments, and
# rows with more than 32 segments (grouped pre-reduction in front of the finish kernel).
LENGTH_CLASSES = [(0, 0), (1, 16), (17, 32), (33, 48), (49, 64), (65, 256), (257, 1024), (1025, 4096), (4097, 131072), (131073, 1 << 40)]
ROW_TOL = 3e-4          # per row, relative to its own norm: the seeded sweep's bar (tests/test_gpu_fuzz.py)
WORST = []              # (config, side, rows compared, entries, worst row, its length) -- printed by the session summary


def stratified_rows(lens, rng, torch, n_rows_target, entry_budget, max_long_len):
    """>= n_rows_target rows (where the side has them), every length class represented: a class gets an equal share of the
    row target, the shares classes cannot fill go to the others in proportion to their size; a class's rows are drawn at
    random, but its LONGEST rows (up to max_long_len, what the oracle finishes in test time) are always in.  The entry
    budget bounds the oracle's work (n_u k^2 fp64 FMAs per row)."""
    picked = []
    classes = []
    for lo, hi in LENGTH_CLASSES:
        idx = torch.nonzero((lens >= lo) & (lens <= min(hi, max_long_len if max_long_len else hi)), as_tuple=False).flatten()
        if idx.numel():
            classes.append((lo, hi, idx))
    share = max(1, n_rows_target // max(1, len(classes)))
    spare = 0
    for lo, hi, idx in classes:
        spare += max(0, share - idx.numel())
    big = sum(idx.numel() for lo, hi, idx in classes if idx.numel() > share)
    per_class = {}
    for lo, hi, idx in classes:
        want = min(idx.numel(), share + (int(spare * idx.numel() / big) if idx.numel() > share and big else 0))
        # entry budget: the classes of long rows may not eat it all
        mean_len = max(1.0, float(lens[idx].double().mean()))
        want = max(1, min(want, int(entry_budget / len(classes) / mean_len)))
        perm = idx[torch.as_tensor(rng.permutation(idx.numel())[:want], device=idx.device)]
        top = idx[torch.topk(lens[idx], min(4, idx.numel())).indices]
        rows = torch.unique(torch.cat([perm, top]))
        per_class[(lo, hi)] = int(rows.numel())
        picked.append(rows)
    return torch.unique(torch.cat(picked)).cpu().numpy().astype(np.int64), per_class


def check_half(core, side, csr, M, G, n_rows, rng, torch, n_sample=100_000, entry_budget=1.2e8, row_offset=0, max_long_len=None,
               config=""):
    """Compare a length-stratified sample of the rows of `side` (>= 100 000 where the side has them: the sweep's strength
    at full-size operand ranges) against the oracle, per row at the sweep's own bar.  M: the opposite factors,
    a host array or -- when the replica is too large to copy (C5's X: 51 GB) -- a device tensor, of which only the rows the
    sample touches are fetched (columns renumbered; a row's system only depends on the rows it references and on G)."""
    import os
    lens = (csr[0][1:] - csr[0][:-1])
    rows, per_class = stratified_rows(lens, rng, torch, min(n_sample, n_rows), entry_budget, max_long_len)
    rp, col, val = sub_csr(csr, rows, torch)
    if isinstance(M, np.ndarray):
        M_host = M
    else:
        used, inv = torch.unique(torch.as_tensor(col, device=M.device).long(), return_inverse=True)
        col = inv.to(torch.int32).cpu().numpy()
        M_host = M[used].cpu().numpy()
    expect = oracle.solve_rows(rp, col, val, M_host, G, threads=max(8, os.cpu_count() or 8))
    got = np.concatenate([core.get_rows(side, rows[i:i + 1_000_000] + row_offset) for i in range(0, len(rows), 1_000_000)])
    assert np.all(np.isfinite(got))
    err = rel(got, expect)
    assert err < REL_TOL, (side, err)
    d = np.linalg.norm(got.astype(np.float64) - expect.astype(np.float64), axis=1)
    nrm = np.maximum(np.linalg.norm(expect.astype(np.float64), axis=1), 1e-30)
    worst_i = int(np.argmax(d / nrm))
    worst = float(d[worst_i] / nrm[worst_i])
    WORST.append((config, "X" if side == pkg.SIDE_X else "Y", len(rows), int(rp[-1]), worst, int(rp[worst_i + 1] - rp[worst_i]), err, per_class))
    assert worst < ROW_TOL, (side, worst, int(rp[worst_i + 1] - rp[worst_i]))
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
        # --- X half (the oracle gets ITS OWN Gramian of the same Y0 wherever it can compute one in test time)
        core.solve_side(pkg.SIDE_X)
        core.check()
        if n_items <= 1_000_000:
            G_for_oracle = oracle.gramian(Y0)
        else:   # not the library's own Gramian: an fp64 matmul on the device (and the two must agree)
            G_for_oracle = independent_gramian(prob["Y0"], torch, dev)
            assert rel(Gy, G_for_oracle) < 5e-7
        # (the C5 shard's rows are compared at full strength by test_c5_rank_at_its_true_shape; here a fifth of it)
        light = dict(n_sample=20_000, entry_budget=2.5e7) if "C5" in name else {}
        max_len_x = check_half(core, pkg.SIDE_X, prob["r_csr"], Y0, G_for_oracle, n_users, rng, torch, config=name, max_long_len=3_000_000, **light)

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
        if n_users <= 1_000_000:
            G_for_oracle = oracle.gramian(X)
        else:
            G_for_oracle = independent_gramian(X, torch, dev)
            assert rel(Gx, G_for_oracle) < 5e-7
        max_len_y = check_half(core, pkg.SIDE_Y, prob["c_csr"], X, G_for_oracle, n_items, rng, torch, config=name, max_long_len=3_000_000, **light)
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


def test_c5_rank_at_its_true_shape():
    """BOTH slices of one rank of C5 at 8 GPUs (ALS:340-389): 12.5M user rows against the 10M x 128 Y
    AND 1.25M item rows (~500 entries each) whose columns index the FULL 100M x 128 X replica (51.2 GB, 200x the
    Infinity Cache) -- the half `c5shard8` never had.  The replicas are allocated at their real size (the HBM budget
    of DESIGN section 3), the rank's rows sit at their real offsets inside them, sampled + longest rows of both halves
    are recomputed by the oracle, and the rows of the other ranks must come out untouched."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1234567890)
    n_users, n_items, nnz_side, k = 12_500_000, 1_250_000, 625_000_000, 128
    prob = bench.rank_problem(torch, synth, n_users, n_items, nnz_side, k, dev, world_emulated=8)
    u_off, i_off = prob["u_off"], prob["i_off"]
    assert prob["X0"].shape == (100_000_000, k) and prob["Y0"].shape == (10_000_000, k)
    assert prob["c_csr"][1].numel() == nnz_side and prob["r_csr"][1].numel() == nnz_side
    assert int(prob["c_csr"][1].max()) > 99_000_000, "item rows must reference the whole 100M-row replica"
    with pkg.ALSCore(k, device=0) as core:
        drv = bench.RankDriver(torch, pkg, core, prob, k, dev)
        X, Y = drv.F[pkg.SIDE_X], drv.F[pkg.SIDE_Y]
        guard_x = (X[u_off - 1].clone(), X[u_off + n_users].clone())
        guard_y = (Y[i_off - 1].clone(), Y[i_off + n_items].clone())
        core.reset_stats()

        # --- user half: rows [u_off, u_off + 12.5M) of X from the 10M-row Y
        drv.half_iteration(pkg.SIDE_X)
        core.check()
        Gy = drv._g.cpu().numpy().copy()
        gs = torch.zeros(k, k, dtype=torch.float64, device=dev)
        core.gramian_partial(pkg.SIDE_Y, 12_345, 1_000_000, gs)
        torch.cuda.synchronize()
        assert rel(gs.cpu().numpy(), oracle.gramian(Y[12_345:1_012_345].cpu().numpy())) < 5e-7
        Gy_indep = independent_gramian(Y, torch, dev)   # the checker's own: an fp64 matmul, not the library's kernels
        assert rel(Gy, Gy_indep) < 5e-7
        check_half(core, pkg.SIDE_X, prob["r_csr"], Y, Gy_indep, n_users, rng, torch, row_offset=u_off, config="C5 rank at its true shape", max_long_len=3_000_000)

        # --- item half: rows [i_off, i_off + 1.25M) of Y from the 100M-row X (the freshly solved user rows included)
        drv.half_iteration(pkg.SIDE_Y)
        core.check()
        Gx = drv._g.cpu().numpy().copy()
        # the Gramian the rank installed = its own partial + the others' (the all-reduce): against the full kernel
        assert rel(Gx, core.gramian(pkg.SIDE_X, fetch=True)) < 2e-7
        core.gramian_partial(pkg.SIDE_X, u_off + 777, 200_000, gs)
        torch.cuda.synchronize()
        assert rel(gs.cpu().numpy(), oracle.gramian(X[u_off + 777:u_off + 200_777].cpu().numpy())) < 5e-7
        Gx_indep = independent_gramian(X, torch, dev)   # 100M x 128: 100 chunks of 1 GB in fp64
        assert rel(Gx, Gx_indep) < 5e-7
        max_len = check_half(core, pkg.SIDE_Y, prob["c_csr"], X, Gx_indep, n_items, rng, torch, row_offset=i_off, config="C5 rank at its true shape", max_long_len=3_000_000)
        assert max_len > 4096, "the popular items of the slice go through the long-row (segments) path"

        st = core.stats()
        assert st["rows_solved"] == n_users + n_items
        assert st["nnz_gathered"] == 2 * nnz_side
        assert st["rows_dual"] > 0
        # nothing outside the rank's slices was written
        assert torch.equal(X[u_off - 1], guard_x[0]) and torch.equal(X[u_off + n_users], guard_x[1])
        assert torch.equal(Y[i_off - 1], guard_y[0]) and torch.equal(Y[i_off + n_items], guard_y[1])
        assert bool(torch.isfinite(Y[i_off:i_off + n_items]).all()) and bool(torch.isfinite(X[u_off:u_off + n_users]).all())
        # HBM budget of a C5 rank: replicas 56.3 GB + both CSR slices 10 GB + scratch, inside 288 GB
        free, total = torch.cuda.mem_get_info()
        assert (total - free) < 200e9, (total - free)
        # ... and the one buffer this slice never asks for: the rotated copy of X (dual path of the item half, +51.2 GB,
        # DESIGN section 3) is only made when the item slice has rows shorter than the feature count and enough of them
        # (none here: the least popular of 10M items still has > 64 of 5e9 entries); a slice that had them would fit too
        assert core.stats()["rows_dual"] == st["rows_dual"]     # all of them user rows: the item half added none
        rotated_copy = torch.empty(100_000_000, k, dtype=torch.float32, device=dev)
        free2, _ = torch.cuda.mem_get_info()
        assert (total - free2) < 0.75 * total, ((total - free2) / 1e9, total / 1e9)
        del rotated_copy


def test_c5_whole_on_one_device():
    """C5 as a WHOLE problem on one MI355X (ALS:340-389): 100M users x 10M items, 5e9 entries, k = 128 --
    80 GB of CSR + CSC, 56 GB of factors, resident together.  The first handle with more than 2^31 entries: every int64
    offset (row_ptr, WorkItem.begin, the work-list counting sort, the segment slots) is exercised at that size.  Both
    half-iterations against the oracle on sampled rows + the longest rows the oracle finishes in test time (2M entries:
    ~490 segments, the grouped pre-reduction of the finish kernel), the Gramians against an independent fp64 matmul."""
    import torch
    dev = torch.device("cuda", 0)
    free, total = torch.cuda.mem_get_info()
    if total < 250e9:
        pytest.skip("needs the 288 GB of an MI355X")
    rng = np.random.default_rng(1234567890)
    n_users, n_items, nnz, k = 100_000_000, 10_000_000, 5_000_000_000, 128
    prob = synth.torch_problem_sliced(n_users, n_items, nnz, k, dev, slices=8)
    assert prob["r_csr"][1].numel() == nnz == prob["c_csr"][1].numel() and nnz > 2 ** 31
    assert int(prob["r_csr"][0][-1]) == nnz and int(prob["c_csr"][0][-1]) == nnz
    # the two orientations hold the same entries: the same multiset of values per item, checked on a few item rows
    for item in (0, 4_999_999, n_items - 1):
        a, b = int(prob["c_csr"][0][item]), int(prob["c_csr"][0][item + 1])
        users = prob["c_csr"][1][a:b].long()
        assert bool((users[1:] > users[:-1]).all())
        u0 = int(users[0]) if b > a else None
        if u0 is not None:
            ra, rb = int(prob["r_csr"][0][u0]), int(prob["r_csr"][0][u0 + 1])
            assert item in prob["r_csr"][1][ra:rb].tolist()
    with pkg.ALSCore(k, device=0) as core:
        core.set_factor_rows(pkg.SIDE_X, n_users)
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_matrix(pkg.SIDE_X, *prob["r_csr"])
        core.set_matrix(pkg.SIDE_Y, *prob["c_csr"])
        Y0 = prob["Y0"].cpu().numpy()
        core.set_factors(pkg.SIDE_Y, Y0)
        core.reset_stats()
        core.half_iteration(pkg.SIDE_X)
        core.check()
        Gy = independent_gramian(prob["Y0"], torch, dev)
        check_half(core, pkg.SIDE_X, prob["r_csr"], Y0, Gy, n_users, rng, torch, n_sample=30_000, entry_budget=4e7, config="C5 whole on one device", max_long_len=2_000_000)
        ptr, n = core.factor_device_ptr(pkg.SIDE_X)

        class _View:
            __cuda_array_interface__ = {"shape": (n, k), "typestr": "<f4", "data": (ptr, False), "version": 2}
        X = torch.as_tensor(_View(), device=dev)
        assert bool(torch.isfinite(X).all())
        core.half_iteration(pkg.SIDE_Y)
        core.check()
        Gx = independent_gramian(X, torch, dev)
        assert rel(core.gramian(pkg.SIDE_X, fetch=True), Gx) < 5e-7
        max_len = check_half(core, pkg.SIDE_Y, prob["c_csr"], X, Gx, n_items, rng, torch, n_sample=30_000, entry_budget=6e7, config="C5 whole on one device", max_long_len=2_000_000)
        assert max_len > 4096 * 32, "the popular items go through the grouped finish of the long-row path"
        st = core.stats()
        assert st["rows_solved"] == n_users + n_items and st["nnz_gathered"] == 2 * nnz
        assert st["rows_dual"] > 0
        free2, _ = torch.cuda.mem_get_info()
        assert (total - free2) < 0.9 * total, ((total - free2) / 1e9, total / 1e9)
