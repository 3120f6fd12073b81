"""The section 8 rows chained the way DelegateGenerationManager does it (DGM:333-355, GL:101-102,
SR:382-441): records -> ingest (row 2) -> factorize (rows a-e) -> reconstruction metric (row 3) ->
model solvers (row 1) -> recommendations (row 4), all on one device, matrices never leaving HBM."""
import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import ingest
from oracle import ingest_oracle as io
from oracle import oracle
from oracle import topn_oracle as to

pytestmark = pytest.mark.gpu


def test_records_to_recommendations():
    rng = np.random.default_rng(2024)
    n_users, n_items, k, n = 3000, 800, 48, 250_000
    # a low-rank preference structure so that the factorization has something to find
    pu, pi = rng.standard_normal((n_users, 6)), rng.standard_normal((n_items, 6))
    u = rng.integers(0, n_users, n)
    i = np.empty(n, np.int64)
    for lo in range(0, n, 20000):                                    # each record: the best of 8 random items for its user
        cand = rng.integers(0, n_items, (min(20000, n - lo), 8))
        s = np.einsum("nf,ncf->nc", pu[u[lo:lo + 20000]], pi[cand])
        i[lo:lo + 20000] = cand[np.arange(len(cand)), s.argmax(1)]
    u_ids, i_ids = u.astype(np.int64) * 7 + 1000, i * 3 + 50         # sparse 64-bit ids
    v = rng.choice([1.0, 2.0, 3.0], n).astype(np.float32)
    v[rng.random(n) < 0.01] = np.nan                                  # a few removes

    (uid, rp, col, val), (iid, cp, ccol, cval) = io.expected_matrices(u_ids, i_ids, v)
    with ingest.Ingest(0) as g, pkg.ALSCore(k) as core:
        g.append(u_ids, i_ids, v)
        g.finish()
        assert np.array_equal(g.ids(pkg.SIDE_X), uid) and np.array_equal(g.ids(pkg.SIDE_Y), iid)
        core.set_factor_rows(pkg.SIDE_X, len(uid))
        core.set_factor_rows(pkg.SIDE_Y, len(iid))
        g.install(core)
        Y0 = (rng.standard_normal((len(iid), k)) / np.sqrt(k)).astype(np.float32)
        core.set_factors(pkg.SIDE_Y, Y0)
        tu = rng.choice(len(uid), 100, replace=False).astype(np.int64)
        ti = rng.choice(len(iid), 100, replace=False).astype(np.int64)
        iters, conv = core.factorize(0.001, 8, False, tu, ti)
        assert 2 <= iters <= 8 and np.isfinite(conv)
        X, Y = core.get_factors(pkg.SIDE_X), core.get_factors(pkg.SIDE_Y)
        # the same call through the oracle (same sample, same start): same iteration count, same factors
        Xo, Yo, iters_o, conv_o = oracle.als_call((rp, col, val), (cp, ccol, cval), len(uid), len(iid), Y0, k,
                                                  conv_threshold=0.001, max_iterations=8, random_y=False,
                                                  test_users=tu, test_items=ti, threads=8)
        assert iters == iters_o
        assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < 1e-4 and np.linalg.norm(Y - Yo) / np.linalg.norm(Yo) < 1e-4
        # reconstruction of the observed entries improved on the first iterate
        s, cnt = core.reconstruction_error()
        so, _ = oracle.reconstruction_error(rp, col, val, X, Y)
        assert cnt == len(col) and abs(s - so) < 1e-9 * max(1.0, so)
        assert s / cnt < 0.9
        # model solvers for fold-in
        solver, norm = core.recompute_solver(pkg.SIDE_Y)
        assert solver.n == k and norm >= 1.0
        # recommendations: never a known item, same as the oracle on the same factors
        users = np.array([0, 17, 1234, len(uid) - 1], np.int64)
        idx, sc, n_out = core.recommend(users, 10)
        for q, uu in enumerate(users):
            known = col[rp[uu]:rp[uu + 1]]
            assert not set(idx[q].tolist()) & set(known.tolist())
            oidx, osc = to.recommend(Y, X[uu], 10, known)
            assert np.allclose(sc[q], osc, rtol=3e-7, atol=1e-12)
            assert len(set(idx[q].tolist()) ^ set(oidx.tolist())) <= 2   # only near-ties may differ


def test_model_file_warm_start(tmp_path):
    """Row 5 on the path: a finished build is saved as model.bin.gz (DGM:270-289), the next build reads
    it back and seeds Y from it (DGM:412-427 -> setPreviousY -> ALS:172-174,304-308): same factors in,
    bit for bit, and the warm build needs fewer iterations than the cold one."""
    from myrrix_recommender_amd import AlternatingLeastSquares, GenerationSerializer, SerializedGeneration
    rng = np.random.default_rng(5)
    n_users, n_items, k = 400, 150, 10
    pu, pi = rng.standard_normal((n_users, 3)), rng.standard_normal((n_items, 3))
    RbyRow, RbyColumn = {}, {}
    for u in range(n_users):
        for i in np.argsort(-(pi @ pu[u]))[:12]:
            RbyRow.setdefault(int(u) + 10_000_000_000, {})[int(i) * 5 - 40] = 1.0 + float(rng.integers(0, 3))
    for u, row in RbyRow.items():
        for i, v in row.items():
            RbyColumn.setdefault(i, {})[u] = v
    cold = AlternatingLeastSquares(RbyRow, RbyColumn, k, 0.001, 40)
    cold.call()
    uid, iid = np.array(list(cold.getX().keys())), np.array(list(cold.getY().keys()))
    ptr = np.concatenate([[0], np.cumsum([len(RbyRow[u]) for u in uid.tolist()])])
    items = np.array([i for u in uid.tolist() for i in RbyRow[u]], np.int64)
    g = SerializedGeneration(userIDs=uid, X=np.stack(list(cold.getX().values())), itemIDs=iid,
                             Y=np.stack(list(cold.getY().values())), knownItemIDs=(uid, ptr, items))
    GenerationSerializer.writeGeneration(g, tmp_path / "model.bin.gz")
    back = GenerationSerializer.readGeneration(tmp_path / "model.bin.gz")
    assert np.array_equal(back.Y, g.Y) and np.array_equal(back.X, g.X) and np.array_equal(back.itemIDs, iid)
    assert {u: set(v.tolist()) for u, v in back.getKnownItemIDs().items()} == {u: set(r) for u, r in RbyRow.items()}
    warm = AlternatingLeastSquares(RbyRow, RbyColumn, k, 0.001, 40)
    warm.setPreviousY(back.getY())
    warm.call()
    assert 1 <= warm.iterations < cold.iterations
    Yc, Yw = np.stack([cold.getY()[i] for i in iid.tolist()]), np.stack([warm.getY()[i] for i in iid.tolist()])
    assert np.linalg.norm(Yw - Yc) / np.linalg.norm(Yc) < 0.05
