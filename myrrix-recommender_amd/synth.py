"""Seeded synthetic interaction matrices (BASELINE.md section 3 / SURVEY.md section 8d).

Item popularity is a power law (popularity rank = floor(I * u^3), i.e. density ~ rank^(-2/3); ranks
are mapped to item ids by a seeded random permutation so that id order carries no popularity
information), user activity is log-normal (sigma 1), strengths are integers 1..5.  Duplicate (user,item) pairs are merged;
`torch_problem` tops the sample up to exactly the requested nnz and plants a low-rank part (see there).  Seed 1234567890 = RandomManager.java:52.
`numpy_problem` is for tests (host arrays), `torch_problem` builds the large bench matrices on the
GPU (CSR by user and CSR by item, int64 row_ptr / int32 col / fp32 val)."""
import numpy as np

SEED = 1234567890


def numpy_problem(n_users, n_items, nnz, k, seed=SEED, negatives=0.0):
    """Returns (r_csr, c_csr, Y0): CSR by user, CSR by item (both (row_ptr, col, val)) and a
    unit-norm random initial Y (plain N(0,1) rows normalised -- not the far-from sampler)."""
    rng = np.random.default_rng(seed)
    act = rng.lognormal(0.0, 1.0, n_users)
    p_user = act / act.sum()
    users = rng.choice(n_users, size=nnz, p=p_user)
    items = np.minimum((n_items * rng.random(nnz) ** 3).astype(np.int64), n_items - 1)
    items = rng.permutation(n_items)[items]
    keys = np.unique(users.astype(np.int64) * n_items + items)
    u = (keys // n_items).astype(np.int64)
    i = (keys % n_items).astype(np.int64)
    vals = rng.integers(1, 6, size=len(keys)).astype(np.float32)
    if negatives > 0:
        vals = np.where(rng.random(len(keys)) < negatives, -vals, vals).astype(np.float32)

    def csr(r, c, v, n):
        order = np.lexsort((c, r))
        r, c, v = r[order], c[order], v[order]
        row_ptr = np.zeros(n + 1, dtype=np.int64)
        np.add.at(row_ptr, r + 1, 1)
        return np.cumsum(row_ptr).astype(np.int64), c.astype(np.int32), v.astype(np.float32)

    r_csr = csr(u, i, vals, n_users)
    c_csr = csr(i, u, vals, n_items)
    Y0 = rng.standard_normal((n_items, k)).astype(np.float32)
    Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True).astype(np.float32)
    return r_csr, c_csr, Y0


PLANT_CLUSTERS = 32   # taste clusters of the planted part
PLANT_CORE = 64       # core items per cluster: item ids [64 c, 64 c + 64)


def torch_problem(n_users, n_items, nnz, k, device, seed=SEED, chunk=1 << 27, planted=0.3):
    """Large problems, generated on `device`.  Returns dict with r_csr, c_csr (torch tensors on the
    device), Y0, the realised nnz (exactly the request unless the matrix cannot hold it) and the
    description of the planted part.

    Planted low-rank part: user u belongs to taste cluster u % 32; a fraction `planted` of the sampled
    interactions goes to the 64 core items of the user's cluster (item ids 64 c .. 64 c + 63, squared
    uniform inside the core), the rest follows the global popularity law.  The (user cluster x core)
    blocks are dense enough (tens of percent) for a rank >= 32 model to predict them, so the
    reconstruction error over those entries says whether ALS is learning; over the long tail nothing is
    predictable at 1e-4 density, which is why the overall mean stays near 1."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    act = torch.exp(torch.randn(n_users, generator=g, device=device, dtype=torch.float32))
    cdf = torch.cumsum(act.double(), 0)
    cdf = (cdf / cdf[-1]).float()
    perm = torch.randperm(n_items, generator=g, device=device)
    n_core = PLANT_CLUSTERS * PLANT_CORE
    if n_items < 4 * n_core:
        planted = 0.0

    def draw(m):
        uu = torch.searchsorted(cdf, torch.rand(m, generator=g, device=device)).clamp_(max=n_users - 1)
        ii = (n_items * torch.rand(m, generator=g, device=device, dtype=torch.float64) ** 3).long().clamp_(max=n_items - 1)
        ii = perm[ii]
        if planted > 0:
            core = (uu % PLANT_CLUSTERS) * PLANT_CORE + (PLANT_CORE * torch.rand(m, generator=g, device=device) ** 2).long().clamp_(max=PLANT_CORE - 1)
            ii = torch.where(torch.rand(m, generator=g, device=device) < planted, core, ii)
        return uu * n_items + ii

    target = min(nnz, n_users * n_items)
    keys = torch.empty(0, dtype=torch.int64, device=device)
    want = target
    for _ in range(64):          # top up until the merged (user, item) pairs reach the request
        parts = [keys]
        done = 0
        while done < want:
            m = min(chunk, want - done)
            parts.append(draw(m))
            done += m
        keys = torch.unique(torch.cat(parts))     # sorted by (user, item)
        del parts
        if keys.numel() >= target:
            break
        want = int((target - keys.numel()) * 1.1) + 1024
    if keys.numel() > target:                     # drop a random surplus: exactly `target` entries
        drop = torch.randperm(keys.numel(), generator=g, device=device)[:keys.numel() - target]
        mask = torch.ones(keys.numel(), dtype=torch.bool, device=device)
        mask[drop] = False
        keys = keys[mask]
        del mask, drop
    n = keys.numel()
    u = torch.div(keys, n_items, rounding_mode="floor")
    i = keys - u * n_items
    del keys
    vals = torch.randint(1, 6, (n,), generator=g, device=device).float()

    def row_ptr_of(r, n_rows):
        counts = torch.bincount(r, minlength=n_rows)
        rp = torch.zeros(n_rows + 1, dtype=torch.int64, device=device)
        torch.cumsum(counts, 0, out=rp[1:])
        return rp

    r_csr = (row_ptr_of(u, n_users), i.int(), vals)
    order = torch.argsort(i * n_users + u)        # by (item, user)
    c_csr = (row_ptr_of(i, n_items), u[order].int(), vals[order])
    del order, u, i
    Y0 = torch.randn(n_items, k, generator=g, device=device, dtype=torch.float32)
    Y0 /= Y0.norm(dim=1, keepdim=True)
    return {"r_csr": r_csr, "c_csr": c_csr, "Y0": Y0, "nnz": n,
            "planted": {"fraction": planted, "clusters": PLANT_CLUSTERS, "core_items": PLANT_CORE} if planted > 0 else None}


def torch_slice(n_rows, n_cols, nnz, device, rows="items", seed=SEED + 1, chunk=1 << 27):
    """One rank's slice of a matrix side as CSR: `n_rows` contiguous rows of R (rows="users": log-normal
    activity over the rows, power-law popularity over the n_cols item columns) or of R^T (rows="items":
    popularity over the rows, activity over the n_cols user columns), exactly `nnz` stored entries, column
    indices over the WHOLE opposite side -- the shape a rank of an 8-GPU C5 run holds (SURVEY.md App. C:
    1.25M item rows x ~500 entries whose columns index a 100M-row X replica).  The activity CDF is kept
    and searched in fp64: at 1e8 columns an fp32 CDF would leave most of them without an entry and shrink
    the set of rows the gather touches.  Returns (row_ptr int64, col int32, val fp32) on `device`."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n_act, n_pop = (n_rows, n_cols) if rows == "users" else (n_cols, n_rows)
    act = torch.exp(torch.randn(n_act, generator=g, device=device, dtype=torch.float32)).double()
    cdf = torch.cumsum(act, 0)
    cdf /= cdf[-1].clone()
    del act
    perm = torch.randperm(n_pop, generator=g, device=device)

    def draw(m):
        a = torch.searchsorted(cdf, torch.rand(m, generator=g, device=device, dtype=torch.float64)).clamp_(max=n_act - 1)
        p = (n_pop * torch.rand(m, generator=g, device=device, dtype=torch.float64) ** 3).long().clamp_(max=n_pop - 1)
        p = perm[p]
        return (a * n_cols + p) if rows == "users" else (p * n_cols + a)

    target = min(nnz, n_rows * n_cols)
    keys = torch.empty(0, dtype=torch.int64, device=device)
    want = target
    for _ in range(64):
        parts = [keys]
        done = 0
        while done < want:
            m = min(chunk, want - done)
            parts.append(draw(m))
            done += m
        keys = torch.unique(torch.cat(parts))     # sorted by (row, column)
        del parts
        if keys.numel() >= target:
            break
        want = int((target - keys.numel()) * 1.1) + 1024
    if keys.numel() > target:
        drop = torch.randperm(keys.numel(), generator=g, device=device)[:keys.numel() - target]
        mask = torch.ones(keys.numel(), dtype=torch.bool, device=device)
        mask[drop] = False
        keys = keys[mask]
        del mask, drop
    del cdf, perm
    r = torch.div(keys, n_cols, rounding_mode="floor")
    c = (keys - r * n_cols).int()
    del keys
    counts = torch.bincount(r, minlength=n_rows)
    del r
    rp = torch.zeros(n_rows + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=rp[1:])
    vals = torch.randint(1, 6, (c.numel(),), generator=g, device=device).float()
    return rp, c, vals


def torch_problem_sliced(n_users, n_items, nnz, k, device, slices=8, seed=SEED):
    """The same kind of problem for sizes whose one-shot generation does not fit next to the problem itself (C5 whole on
    one GPU: 100M x 10M, 5e9 entries -- 40 GB per orientation): R is generated `slices` contiguous user ranges at a time
    (torch_slice, rows="users", one seed per range) straight into preallocated CSR arrays, and R^T is built from it by a
    counting sort over the items -- per range a stable sort by item, scattered behind the entries the earlier ranges left
    for that item, so that every item row is ascending by user like torch_problem's.  Both orientations hold exactly the
    same `nnz` entries (every int64 offset beyond 2^31 is exercised from 2.15e9 entries on).  Nothing is planted."""
    import torch
    assert n_users % slices == 0 and nnz % slices == 0
    per_u, per_nnz = n_users // slices, nnz // slices
    r_rp = torch.zeros(n_users + 1, dtype=torch.int64, device=device)
    r_col = torch.empty(nnz, dtype=torch.int32, device=device)
    r_val = torch.empty(nnz, dtype=torch.float32, device=device)
    item_count = torch.zeros(n_items, dtype=torch.int64, device=device)
    for s_ in range(slices):
        rp, col, val = torch_slice(per_u, n_items, per_nnz, device, rows="users", seed=seed + 17 * (s_ + 1))
        assert int(rp[-1]) == per_nnz
        r_rp[s_ * per_u + 1:(s_ + 1) * per_u + 1] = rp[1:] + s_ * per_nnz
        r_col[s_ * per_nnz:(s_ + 1) * per_nnz] = col
        r_val[s_ * per_nnz:(s_ + 1) * per_nnz] = val
        item_count += torch.bincount(col, minlength=n_items)
        del rp, col, val
        torch.cuda.empty_cache()
    c_rp = torch.zeros(n_items + 1, dtype=torch.int64, device=device)
    torch.cumsum(item_count, 0, out=c_rp[1:])
    del item_count
    c_col = torch.empty(nnz, dtype=torch.int32, device=device)
    c_val = torch.empty(nnz, dtype=torch.float32, device=device)
    filled = torch.zeros(n_items, dtype=torch.int64, device=device)
    for s_ in range(slices):
        lo, hi = s_ * per_nnz, (s_ + 1) * per_nnz
        items = r_col[lo:hi].long()
        lens = r_rp[s_ * per_u + 1:(s_ + 1) * per_u + 1] - r_rp[s_ * per_u:(s_ + 1) * per_u]
        users = torch.repeat_interleave(torch.arange(s_ * per_u, (s_ + 1) * per_u, device=device, dtype=torch.int32), lens)
        del lens
        order = torch.argsort(items, stable=True)            # by item, users ascending inside an item
        items = items[order]
        cnt = torch.bincount(items, minlength=n_items)
        start = torch.cumsum(cnt, 0) - cnt                      # first position of every item inside this range's sorted run
        pos = c_rp[:-1][items] + filled[items] + (torch.arange(per_nnz, device=device) - start[items])
        del start, items
        c_col[pos] = users[order]
        c_val[pos] = r_val[lo:hi][order]
        filled += cnt
        del pos, order, users, cnt
        torch.cuda.empty_cache()
    del filled
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    Y0 = torch.randn(n_items, k, generator=g, device=device, dtype=torch.float32)
    Y0 /= Y0.norm(dim=1, keepdim=True)
    return {"r_csr": (r_rp, r_col, r_val), "c_csr": (c_rp, c_col, c_val), "Y0": Y0, "nnz": nnz, "planted": None}


def planted_reconstruction_error(prob, X, Y, sample=4_000_000):
    """Mean of max(0, 1 - x_u . y_i) over (a sample of) the stored entries of the planted part: core items
    of the user's own cluster.  torch on the device; diagnostics only."""
    import torch
    if not prob.get("planted"):
        return None
    rp, col, _ = prob["r_csr"]
    n_core = PLANT_CLUSTERS * PLANT_CORE
    idx = torch.nonzero(col < n_core).squeeze(1)
    if idx.numel() > sample:
        idx = idx[torch.randperm(idx.numel(), device=idx.device)[:sample]]
    users = torch.searchsorted(rp, idx, right=True) - 1
    items = col[idx].long()
    own = (items // PLANT_CORE) == (users % PLANT_CLUSTERS)
    users, items = users[own], items[own]
    if users.numel() == 0:
        return None
    err = 0.0
    for s in range(0, users.numel(), 1 << 20):
        d = (X[users[s:s + (1 << 20)]].double() * Y[items[s:s + (1 << 20)]].double()).sum(1)
        err += torch.clamp(1.0 - d, min=0.0).sum().item()
    return {"mean": err / users.numel(), "entries_sampled": int(users.numel())}
