"""SURVEY.md section 8(f) row 2 meets 8(e): InputFilesReader.readInputFiles (IFR:64-211) -> the GROUP that factorizes
(DelegateGenerationManager.java:406-410), device to device -- mals_ingest_install_group cuts both CSRs at the group's
cost-balanced bounds and hands every member its slices, its users' knownItemIDs and the userTagIDs mask.  The GPU box has one
device: N members on device 0 (slices borrowed in place), and the same with MALS_INSTALL_COPY (every member copies its slices -- what a
member on another device always does -- and the ingest is destroyed before the first iteration)."""
import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import _lib, ingest
from oracle import ingest_text_oracle as to
from oracle import oracle, topn_oracle

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def corpus(seed, n_users, n_items, n_lines, tags=True):
    """A reference-shaped input file: numeric lines (duplicates sum, some removed again, some near zero and pruned) plus
    tag lines of both kinds (IFR:114-130,159-165)."""
    rng = np.random.default_rng(seed)
    lines = ["user,item,value"]
    for _ in range(n_lines):
        u = int(rng.integers(0, n_users)) * 7 + 1000
        i = int(min(n_items - 1, rng.zipf(1.3) - 1 if rng.random() < 0.5 else rng.integers(0, n_items))) * 3 + 50
        r = rng.random()
        if r < 0.03:
            lines.append("%d,%d," % (u, i))                     # remove
        elif r < 0.08:
            lines.append("%d,%d,0.00001" % (u, i))              # pruned by removeSmall, stays a known item
        elif r < 0.12:
            lines.append("%d,%d" % (u, i))                      # value absent = 1
        else:
            lines.append("%d,%d,%s" % (u, i, rng.choice(["1", "2", "3.5", "-1", "0.5"])))
    if tags:
        for t in range(12):                                     # user tags: a tag in the ITEM column -> userTagIDs, a pseudo-item row
            for _ in range(int(rng.integers(1, 30))):
                lines.append('%d,"genre%d",%s' % (int(rng.integers(0, n_users)) * 7 + 1000, t, rng.choice(["1", "2"])))
        for t in range(5):                                      # item tags: a tag in the USER column -> itemTagIDs, a pseudo-user row
            for _ in range(int(rng.integers(1, 20))):
                lines.append('"crowd%d",%d,1' % (t, int(rng.integers(0, n_items)) * 3 + 50))
        lines.append('1000,"gone",1')                           # a user tag whose only entry is removed again: owns no row
        lines.append('1000,"gone",')
    rng.shuffle(lines[1:])
    return ("\n".join(lines) + "\n").encode()


@pytest.mark.parametrize("world,k,force_copy", [(3, 64, False), (3, 64, True), (2, 30, True), (4, 128, False)])
def test_text_to_group_two_iterations_match_the_oracle(world, k, force_copy):
    data = corpus(40 + world, 1500, 400, 60000)
    want = to.expected([data])
    (uid, rp, col, val), (iid, cp, ccol, cval) = want["csr_x"], want["csr_y"]
    Y0 = (np.random.default_rng(k).standard_normal((len(iid), k)) / np.sqrt(k)).astype(np.float32)
    with ingest.Ingest(0) as g, pkg.GroupALS.single_process(k, [0] * world, backend=_lib.GROUP_PEER_COPY, exchange_chunks=3) as grp:
        g.set_option(_lib.INGEST_OPT_KNOWN_ITEMS, 1)
        g.append_text(data, True)
        g.finish()
        ingest_tag_items = g.tag_items()
        g.install_group(grp, copy=force_copy)                   # declares the replicas from the ingest's counts
        if force_copy:
            g.close()                                           # MALS_INSTALL_COPY: nothing of the ingest is needed any more
        bx, by = grp.bounds(pkg.SIDE_X), grp.bounds(pkg.SIDE_Y)
        assert bx[0] == 0 and bx[-1] == len(uid) and by[-1] == len(iid) and np.all(np.diff(bx) >= 0) and np.all(np.diff(by) >= 0)
        # the slices are cost-balanced like any other upload: mals_plan_shards on the same row pointers
        assert np.array_equal(bx, pkg.group.plan_shards(rp, world, k)) and np.array_equal(by, pkg.group.plan_shards(cp, world, k))
        grp.set_factors(pkg.SIDE_Y, Y0)
        grp.iterate(2)
        X = grp.get_factors(pkg.SIDE_X, 0, len(uid))
        Y = grp.get_factors(pkg.SIDE_Y, 0, len(iid))
        # serving from the members: a user is answered by the member that holds its row -- its knownItemIDs (entries that
        # removeSmall pruned included) skipped, no userTagID ever returned
        tags = want["user_tag_ids"]
        tag_idx = np.searchsorted(iid, tags)
        tag_idx = tag_idx[(tag_idx < len(iid)) & (iid[np.minimum(tag_idx, len(iid) - 1)] == tags)]
        assert len(tag_idx) == 12 and len(tags) == 13           # "gone" owns no row
        assert np.array_equal(np.sort(ingest_tag_items[ingest_tag_items >= 0]), np.sort(tag_idx)) and (ingest_tag_items < 0).sum() == 1
        for i in range(world):
            core, rank = grp.local(i)
            assert core.tag_item_count() == len(tag_idx)
            users = np.arange(bx[rank], bx[rank + 1], dtype=np.int64)[:40]
            if len(users) == 0:
                continue
            idx, sc, cnt = core.recommend(users, 8)
            for q, u in enumerate(users):
                known = want["known_idx"][want["known_ptr"][u]:want["known_ptr"][u + 1]]
                oidx, osc = topn_oracle.recommend(Y, X[u], 8, known, tag_idx)
                assert cnt[q] == len(oidx) and np.array_equal(idx[q, :cnt[q]], oidx), (rank, u)
                assert np.array_equal(sc[q, :cnt[q]].view(np.uint32), np.asarray(osc, np.float32).view(np.uint32))
            # the same users through the group's router: answered by whichever member holds them
            r_idx, r_sc, r_cnt = grp.recommend(users, 8)
            assert np.array_equal(r_idx, idx) and np.array_equal(r_sc.view(np.uint32), sc.view(np.uint32)) and np.array_equal(r_cnt, cnt)
            # a user of ANOTHER member's slice: this member does not hold its known items and says so
            other = int(bx[(rank + 1) % world]) if world > 1 and bx[(rank + 1) % world] < len(uid) and not (bx[rank] <= bx[(rank + 1) % world] < bx[rank + 1]) else None
            if other is not None:
                with pytest.raises(pkg.MalsError):
                    core.recommend(np.array([other], np.int64), 8)
        # users of ALL slices in one call, in mixed order: the router cuts the call into runs per owner
        mixed = np.random.default_rng(world).permutation(len(uid))[:50].astype(np.int64)
        m_idx, m_sc, m_cnt = grp.recommend(mixed, 8)
        for q, u in enumerate(mixed):
            known = want["known_idx"][want["known_ptr"][u]:want["known_ptr"][u + 1]]
            oidx, osc = topn_oracle.recommend(Y, X[u], 8, known, tag_idx)
            assert m_cnt[q] == len(oidx) and np.array_equal(m_idx[q, :m_cnt[q]], oidx)
    Xo, Yo = None, Y0
    for _ in range(2):
        Xo = oracle.half_iteration(rp, col, val, Yo, threads=4)
        Yo = oracle.half_iteration(cp, ccol, cval, Xo, threads=4)
    assert rel(X, Xo) < REL_TOL and rel(Y, Yo) < REL_TOL, (rel(X, Xo), rel(Y, Yo))


def test_install_group_equals_the_upload_through_the_host():
    """The same input through mals_ingest_install_group and through get_csr + mals_group_set_matrix: bitwise the same factors."""
    k, world = 50, 3
    data = corpus(7, 900, 300, 30000, tags=False)
    res = []
    for direct in (True, False):
        with ingest.Ingest(0) as g, pkg.GroupALS.single_process(k, [0] * world, backend=_lib.GROUP_PEER_COPY) as grp:
            g.append_text(data, True)
            g.finish()
            c = g.counts()
            Y0 = (np.random.default_rng(1).standard_normal((c["items"], k)) / np.sqrt(k)).astype(np.float32)
            if direct:
                g.install_group(grp)
            else:
                grp.set_factor_rows(pkg.SIDE_X, c["users"])
                grp.set_factor_rows(pkg.SIDE_Y, c["items"])
                grp.set_matrix(pkg.SIDE_X, *g.csr(pkg.SIDE_X))
                grp.set_matrix(pkg.SIDE_Y, *g.csr(pkg.SIDE_Y))
            grp.set_factors(pkg.SIDE_Y, Y0)
            grp.iterate(2)
            res.append((grp.get_factors(pkg.SIDE_X, 0, c["users"]), grp.get_factors(pkg.SIDE_Y, 0, c["items"])))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_tagged_corpus_end_to_end_on_one_handle():
    """text -> mals_ingest_install -> mals_recommend on a tagged input: the reference never returns a userTagID
    (RecommendIterator.java:72); before round 6 the tag pseudo-items came back."""
    k = 16
    data = corpus(99, 300, 5000, 20000)
    want = to.expected([data])
    (uid, rp, col, val), (iid, cp, ccol, cval) = want["csr_x"], want["csr_y"]
    tags = want["user_tag_ids"]
    pos = np.searchsorted(iid, tags)
    tag_idx = pos[(pos < len(iid)) & (iid[np.minimum(pos, len(iid) - 1)] == tags)]
    rng = np.random.default_rng(3)
    X = rng.standard_normal((len(uid), k)).astype(np.float32)
    Y = rng.standard_normal((len(iid), k)).astype(np.float32)
    Y[tag_idx] *= 4.0                                           # make the tag rows the best-scoring rows for many users
    with ingest.Ingest(0) as g, pkg.ALSCore(k) as core:
        g.set_option(_lib.INGEST_OPT_KNOWN_ITEMS, 1)
        g.append_text(data, True)
        g.finish()
        g.install(core)                                         # declares Y's rows itself when there are tags to strike
        core.set_factor_rows(pkg.SIDE_X, len(uid))
        core.set_factors(pkg.SIDE_X, X)
        core.set_factors(pkg.SIDE_Y, Y)
        assert core.tag_item_count() == len(tag_idx) > 0
        users = np.arange(len(uid), dtype=np.int64)
        idx, sc, cnt = core.recommend(users, 6)
        hit_without_mask = 0
        for q in range(len(uid)):
            known = want["known_idx"][want["known_ptr"][q]:want["known_ptr"][q + 1]]
            oidx, osc = topn_oracle.recommend(Y, X[q], 6, known, tag_idx)
            assert np.array_equal(idx[q], oidx) and np.array_equal(sc[q].view(np.uint32), np.asarray(osc, np.float32).view(np.uint32))
            hit_without_mask += bool(set(topn_oracle.recommend(Y, X[q], 6, known)[0].tolist()) & set(tag_idx.tolist()))
        assert hit_without_mask > 0                             # the mask mattered on this input
        # an install from an ingest WITHOUT known items drops the ones of the earlier install
        with ingest.Ingest(0) as g2:
            g2.append_text(data, True)
            g2.finish()
            g2.install(core)
            core.set_factors(pkg.SIDE_X, X)
            idx2, sc2, _ = core.recommend(users[:20], 6)
            for q in range(20):
                oidx, osc = topn_oracle.recommend(Y, X[q], 6, col[rp[q]:rp[q + 1]], tag_idx)
                assert np.array_equal(idx2[q], oidx)
