"""SURVEY.md section 8(f) row 2, the text half: mals_ingest_append_text / _read_file / _read_dir (bytes of the input
files -> records -> CSR, on the device) against oracle/ingest_text_oracle.py's restatement of
InputFilesReader.readInputFiles.  Everything is integer / byte / index work or single fp32 adds in file order:
the bar is bit-exact -- line counts, record arrays, ids, CSR, values, tag id sets, knownItemIDs."""
import gzip
import os
import zipfile

import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import _lib, ingest
from oracle import ingest_text_oracle as to
from tests import text_corpus as tc

pytestmark = pytest.mark.gpu


def build_corpus(seed, n_lines, p_odd, bad_budget=90, terminators=("\n",), first_line=None, n_users=50, n_items=40, final_newline=True):
    """Lines from tests/text_corpus.py, minus the ones that would end the read (fatal tokens, more than `bad_budget`
    bad lines) -- those have their own tests."""
    rng = np.random.default_rng(seed)
    out = bytearray()
    bad = 0
    k = 0
    while k < n_lines:
        line = first_line if (k == 0 and first_line is not None) else tc.line_bytes(rng, n_users, n_items, p_odd)
        st = to.parse_line(to.java_utf8_decode(line), k + 1)[0]
        if st == to.FATAL:
            continue
        if st == to.BAD:
            if bad >= bad_budget:
                continue
            bad += 1
        out += line
        k += 1
        if k < n_lines or final_newline:
            out += str(rng.choice(list(terminators))).encode()
    return bytes(out)


def check(streams, thr=1.0e-4, block_bytes=None, pieces=None, known=True, part=None):
    """streams: list of bytes (files in order).  pieces: how each file is cut for append_text (None = whole).  part:
    MALS_INGEST_OPT_PARTITION_RECORDS (the finish then runs user-id range by user-id range, ingest_big_host.h)."""
    want = to.expected(streams, thr)
    with ingest.Ingest(0, thr) as g:
        if part:
            g.set_option(_lib.INGEST_OPT_PARTITION_RECORDS, part)
        if known:
            g.set_option(_lib.INGEST_OPT_KNOWN_ITEMS, 1)
        if block_bytes:
            g.set_option(_lib.INGEST_OPT_TEXT_BLOCK_BYTES, block_bytes)
        rng = np.random.default_rng(len(streams))
        for data in streams:
            if pieces is None:
                g.append_text(data, True)
            else:
                cuts = sorted(set(rng.integers(0, len(data) + 1, pieces).tolist())) if len(data) else []
                prev = 0
                for c in cuts:
                    g.append_text(data[prev:c], False)
                    prev = c
                g.append_text(data[prev:], True)
        info = g.text_info()
        assert info["lines"] == want["lines"] and info["bad_lines"] == want["bad_lines"], (info, want["lines"], want["bad_lines"])
        assert info["records"] == len(want["users"])
        assert info["header_lines"] == int((want["statuses"] == to.HEADER).sum())
        assert info["skipped_lines"] == int((want["statuses"] == to.SKIP).sum())
        g.finish()
        c = g.counts()
        (uid, rp, col, val), (iid, cp, ccol, cval) = want["csr_x"], want["csr_y"]
        assert c == {"records": len(want["users"]), "users": len(uid), "items": len(iid), "nnz": len(col)}
        assert np.array_equal(g.ids(pkg.SIDE_X), uid) and np.array_equal(g.ids(pkg.SIDE_Y), iid)
        grp, gcol, gval = g.csr(pkg.SIDE_X)
        assert np.array_equal(grp, rp) and np.array_equal(gcol, col)
        assert np.array_equal(gval.view(np.uint32), val.view(np.uint32))
        gcp, gccol, gcval = g.csr(pkg.SIDE_Y)
        assert np.array_equal(gcp, cp) and np.array_equal(gccol, ccol)
        assert np.array_equal(gcval.view(np.uint32), cval.view(np.uint32))
        assert np.array_equal(g.tag_ids(_lib.ITEM_TAG_IDS), want["item_tag_ids"])
        assert np.array_equal(g.tag_ids(_lib.USER_TAG_IDS), want["user_tag_ids"])
        if known:
            kp, ki = g.known_items()
            assert np.array_equal(kp, want["known_ptr"]) and np.array_equal(ki, want["known_idx"])
        return g.text_info()


def test_reference_shaped_file():
    data = b"user,item,value\n1,10,1\n1,11,2.5\n2,10,0.00001\n1,10,\n3,12,1\n3,12,\n\"t\",10,2\n4,\"s\",1\n#comment\n\n5,13\n"
    info = check([data])
    assert info["header_lines"] == 1 and info["bad_lines"] == 0 and info["records"] == 9


# MALS_TEXT_SEEDS=N: N more corpora of 2000-5000 lines with every share of odd lines (a longer hunt; profiles/r5_parity_evidence.txt)
_MORE_TEXT = [(1000 + i, 2000 + 37 * (i % 83), [0.05, 0.25, 0.5, 0.75, 0.95][i % 5]) for i in range(int(os.environ.get("MALS_TEXT_SEEDS", "0")))]


@pytest.mark.parametrize("seed,n_lines,p_odd", [(1, 1, 0.5), (2, 63, 0.5), (3, 64, 0.3), (4, 257, 0.6), (5, 5000, 0.25), (6, 40000, 0.1),
                                                (7, 3000, 0.9), (8, 20000, 0.5)] + _MORE_TEXT)
def test_fuzzed_corpus_matches_oracle(seed, n_lines, p_odd):
    info = check([build_corpus(seed, n_lines, p_odd)])
    if p_odd >= 0.25 and n_lines >= 3000:
        assert info["full_parser_lines"] > 0


def test_plain_numeric_file_stays_on_the_fast_parser():
    rng = np.random.default_rng(11)
    vals = ["1", "2.5", "-1", "0.00003", "4.25e0", " 3 ", "\t5", "1e-2", "+2", "0.1", "3.4028235e38", ""]
    lines = ["%d,%d,%s" % (rng.integers(0, 3000), rng.integers(0, 2000) + 5000, vals[int(rng.integers(0, len(vals)))]) for _ in range(30000)]
    lines += ["%d , %d" % (rng.integers(0, 3000), rng.integers(0, 2000) + 5000) for _ in range(2000)]
    info = check([("\n".join(lines) + "\n").encode()])
    assert info["full_parser_lines"] == 0 and info["bad_lines"] == 0 and info["records"] == 32000


@pytest.mark.parametrize("terms", [("\r\n",), ("\r",), ("\n", "\r\n", "\r"), ("\n", "\n\n", "\r\r", "\r\n\r\n")])
def test_line_terminators(terms):
    check([build_corpus(21, 4000, 0.3, terminators=terms)])
    check([build_corpus(22, 4000, 0.3, terminators=terms, final_newline=False)])


@pytest.mark.parametrize("block", [1, 7, 64, 4096, 100000])
def test_any_block_size_gives_the_same_result(block):
    """A block boundary may fall anywhere: inside a line, inside a multi-byte char, between '\\r' and '\\n'."""
    n = 300 if block < 64 else 6000
    check([build_corpus(31, n, 0.4, terminators=("\n", "\r\n", "\r"))], block_bytes=block)


def test_any_split_of_the_caller_gives_the_same_result():
    data = build_corpus(41, 5000, 0.4, terminators=("\n", "\r\n", "\r"))
    check([data], pieces=40)
    check([data], pieces=400, block_bytes=512)
    # split exactly between '\r' and '\n'
    d2 = b"1,2,3\r\n4,5,6\r\n7,8\r9,9\r\n"
    want = to.expected([d2])
    for cut in range(len(d2) + 1):
        with ingest.Ingest(0) as g:
            g.append_text(d2[:cut], False)
            g.append_text(d2[cut:], True)
            assert g.text_info()["lines"] == want["lines"] == 4
            g.finish()
            assert g.counts()["nnz"] == len(want["csr_x"][2])


def test_several_files_one_line_counter():
    """IFR:92-99: `lines` and `badLines` run over all files -- only the very first line of the first file can be a
    header; a file's unterminated last line does not join the next file's first."""
    a = b"user,item\n1,2,3\n7,8"            # no newline at the end
    b = b"user,item\n4,5,6\n"              # this 'header' is line 4: a bad line
    info = check([a, b, b""])
    assert info["lines"] == 5 and info["header_lines"] == 1 and info["bad_lines"] == 1 and info["records"] == 3
    check([build_corpus(51, 2000, 0.3, bad_budget=30), build_corpus(52, 2000, 0.3, bad_budget=30, final_newline=False),
           build_corpus(53, 2000, 0.3, bad_budget=30)])


def test_too_many_bad_lines():
    ok, bad = b"1,2,3\n", b"x\n"
    check([ok + bad * 101])                                   # the 101st bad line is the last line: no line follows, no throw
    for streams in ([ok + bad * 101 + b"\n"], [ok + bad * 101, b"#\n"], [ok + bad * 60, bad * 60]):
        with pytest.raises(to.TooManyBadLines):
            to.read_streams(streams)
        with ingest.Ingest(0) as g:
            with pytest.raises(pkg.MalsError) as e:
                for s in streams:
                    g.append_text(s, True)
            assert e.value.status == _lib.IO_ERROR and "Too many bad lines" in str(e.value)
            with pytest.raises(pkg.MalsError):                # the ingest stays failed, like the IOException ends the read
                g.finish()
    # across blocks: the counter is carried
    with ingest.Ingest(0) as g:
        g.set_option(_lib.INGEST_OPT_TEXT_BLOCK_BYTES, 16)
        with pytest.raises(pkg.MalsError):
            g.append_text(ok + bad * 150, True)
    assert check([b"x\n" + bad * 100])["bad_lines"] == 100     # line 1 is a header, not a bad line


def test_lone_quote_token_is_fatal_like_the_reference():
    for data in (b"1,2,3\n\",5,1\n", b"1,\"\n", b"1,2\n3, \" ,1\n"):
        with pytest.raises(to.UncaughtStringIndexOutOfBounds):
            to.read_streams([data])
        with ingest.Ingest(0) as g:
            with pytest.raises(pkg.MalsError) as e:
                g.append_text(data, True)
            assert e.value.status == _lib.INVALID_ARG
    check([b"x,\"\n1,2\n"])                                   # the user token fails first: a header, the quote is never looked at


def test_text_on_the_device_and_mixed_with_records():
    import torch
    data = build_corpus(61, 8000, 0.2)
    want = to.expected([data])
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    with ingest.Ingest(0) as g:
        g.set_option(_lib.INGEST_OPT_TEXT_BLOCK_BYTES, 50000)
        g.append_text(t, True)
        g.finish()
        assert np.array_equal(g.csr(pkg.SIDE_X)[2].view(np.uint32), want["csr_x"][3].view(np.uint32))
        assert np.array_equal(g.csr(pkg.SIDE_X)[1], want["csr_x"][2])
    # records appended directly keep their place in the stream
    r = to.read_streams([data])
    from oracle import ingest_oracle as io
    extra_u, extra_i, extra_v = np.array([1, 2], np.int64), np.array([1001, 1002], np.int64), np.array([2.0, np.nan], np.float32)
    wu = np.concatenate([extra_u, r["users"]])
    wi = np.concatenate([extra_i, r["items"]])
    wv = np.concatenate([extra_v, r["values"]])
    (uid, rp, col, val), _ = io.expected_matrices(wu, wi, wv)
    with ingest.Ingest(0) as g:
        g.append(extra_u, extra_i, extra_v)
        g.append_text(data, True)
        g.finish()
        grp, gcol, gval = g.csr(pkg.SIDE_X)
        assert np.array_equal(grp, rp) and np.array_equal(gcol, col) and np.array_equal(gval.view(np.uint32), val.view(np.uint32))


def test_read_dir_like_the_reference(tmp_path):
    """IFR:71-86 + FileLineIterator.java:92-102: the name filter, last-modified order, .gz inflated, .zip read as empty."""
    d = tmp_path / "in"
    d.mkdir()
    f1 = build_corpus(71, 3000, 0.3, bad_budget=20)
    f2 = build_corpus(72, 3000, 0.3, bad_budget=20, final_newline=False)
    f3 = build_corpus(73, 50000, 0.05, bad_budget=20, n_users=2000, n_items=900)
    (d / "b.csv").write_bytes(f1)
    with gzip.open(d / "a.csv.gz", "wb") as f:
        f.write(f2)
    with gzip.open(d / "c.csv.gz", "wb") as f:                 # two members, like `cat x.gz y.gz`
        f.write(f3[:70000])
    with gzip.open(d / "c.csv.gz", "ab") as f:
        f.write(f3[70000:])
    with zipfile.ZipFile(d / "z.csv.zip", "w") as z:
        z.writestr("inner.csv", "9999,9999,9\n")
    (d / "notes.txt").write_bytes(b"1,2,3\n")
    (d / ".csv").write_bytes(b"1,2,3\n")                        # ".+\\.csv" needs a char before the dot
    (d / "x.csv.bak").write_bytes(b"1,2,3\n")
    order = ["c.csv.gz", "z.csv.zip", "b.csv", "a.csv.gz"]
    for k, name in enumerate(order):
        os.utime(d / name, (1_600_000_000 + 10 * k, 1_600_000_000 + 10 * k))
    assert [os.path.basename(p) for p in to.list_input_files(str(d))] == order
    want = to.read_input_files(str(d))
    assert want["lines"] > 50000
    got = ingest.readInputFiles(str(d))
    uid, rp, col, val = to.expected([to.file_bytes(p) for p in to.list_input_files(str(d))])["csr_x"]
    assert got["info"]["lines"] == want["lines"] and got["info"]["bad_lines"] == want["bad_lines"]
    assert np.array_equal(got["user_ids"], uid)
    assert np.array_equal(got["RbyRow"][0], rp) and np.array_equal(got["RbyRow"][1], col)
    assert np.array_equal(got["RbyRow"][2].view(np.uint32), val.view(np.uint32))
    assert np.array_equal(got["itemTagIDs"], want["item_tag_ids"]) and np.array_equal(got["userTagIDs"], want["user_tag_ids"])
    assert 9999 not in got["user_ids"]                          # nothing comes out of the .zip
    # a missing directory reads nothing (IFR:80-83); a gzip name over plain bytes is an IOException
    assert ingest.readInputFiles(str(tmp_path / "nope"))["info"]["lines"] == 0
    (d / "bad.csv.gz").write_bytes(b"1,2,3\n")
    with pytest.raises(pkg.MalsError) as e:
        ingest.readInputFiles(str(d))
    assert e.value.status == _lib.IO_ERROR


def test_ingest_from_text_feeds_the_factorizer():
    """readInputFiles -> runFactorization (DelegateGenerationManager.java:333-355) without the matrices leaving HBM."""
    rng = np.random.default_rng(3)
    n_users, n_items, k = 400, 150, 24
    u = rng.integers(0, n_users, 20000) + 1000
    i = rng.integers(0, n_items, 20000) + 50
    v = rng.choice([1, 2, 3], 20000)
    data = "".join("%d,%d,%d\n" % t for t in zip(u, i, v)).encode()
    want = to.expected([data])
    (uid, rp, col, val), (iid, cp, ccol, cval) = want["csr_x"], want["csr_y"]
    from oracle import oracle
    Y0 = (rng.standard_normal((len(iid), k)) / np.sqrt(k)).astype(np.float32)
    with ingest.Ingest(0) as g, pkg.ALSCore(k) as core:
        g.append_text(data, True)
        g.finish()
        core.set_factor_rows(pkg.SIDE_X, len(uid))
        core.set_factor_rows(pkg.SIDE_Y, len(iid))
        g.install(core)
        core.set_factors(pkg.SIDE_Y, Y0)
        core.half_iteration(pkg.SIDE_X)
        X = core.get_factors(pkg.SIDE_X)
    Xo = oracle.half_iteration(rp, col, val, Y0)
    assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < 1e-4


def test_big_plain_file_round_trip():
    """2M lines: values and ids survive the trip through text exactly (a size-independent property: printing a
    float32 with 9 significant digits and parsing it back is the identity)."""
    rng = np.random.default_rng(77)
    n = 2_000_000
    u = rng.integers(0, 300000, n)
    i = rng.integers(0, 50000, n)
    v = (rng.standard_normal(n) * 3).astype(np.float32)
    v[np.abs(v) < 1e-3] = 1.0
    lines = np.char.add(np.char.add(np.char.add(u.astype(str), ","), np.char.add(i.astype(str), ",")), np.char.mod("%.9g", v))
    data = ("\n".join(lines.tolist()) + "\n").encode()
    from oracle import ingest_oracle as io
    with ingest.Ingest(0) as a, ingest.Ingest(0) as b:
        a.set_option(_lib.INGEST_OPT_TEXT_BLOCK_BYTES, 8 << 20)
        a.append_text(data, True)
        info = a.text_info()
        assert info["lines"] == n and info["records"] == n and info["bad_lines"] == 0 and info["full_parser_lines"] == 0
        b.append(u.astype(np.int64), i.astype(np.int64), v)
        a.finish()
        b.finish()
        for side in (pkg.SIDE_X, pkg.SIDE_Y):
            ca, cb = a.csr(side), b.csr(side)
            assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1])
            assert np.array_equal(ca[2].view(np.uint32), cb[2].view(np.uint32))
        assert np.array_equal(a.ids(pkg.SIDE_X), b.ids(pkg.SIDE_X))


def test_a_hundred_million_lines_equal_their_records():
    """1e8 lines formatted on the device (with removes), in pieces split inside lines: the text path leaves exactly the
    matrices, ids and counts the record path leaves for the same (user, item, value) stream -- the size-independent property
    of this row (the pure-Python oracle replays 4e4 lines a second)."""
    import torch
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_ingest", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_ingest.py"))
    bi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bi)
    n, n_users, n_items = 100_000_000, 5_000_000, 600_000
    gen = torch.Generator(device="cuda").manual_seed(99)
    u = torch.randint(0, n_users, (n,), device="cuda", generator=gen)
    i = (torch.rand(n, device="cuda", generator=gen).pow_(2.0) * n_items).long().clamp_(max=n_items - 1)
    v = torch.randint(1, 6, (n,), device="cuda", generator=gen).float()
    v[torch.rand(n, device="cuda", generator=gen) < 0.02] = float("nan")           # "user,item," = remove (IFR:137,165-167)
    with ingest.Ingest(0) as a, ingest.Ingest(0) as b:
        piece = 12_500_000
        carry = None
        for lo in range(0, n, piece):
            t = bi.format_lines(torch, u[lo:lo + piece], i[lo:lo + piece], v[lo:lo + piece])
            if carry is not None:
                t = torch.cat([carry, t])
            cut = t.numel() - 7 if lo + piece < n else t.numel()                   # hand over all but 7 bytes: a split inside a line
            a.append_text(t[:cut], end_of_file=lo + piece >= n)
            carry = t[cut:].clone() if cut < t.numel() else None
            del t
        info = a.text_info()
        assert info["lines"] == n and info["records"] == n and info["bad_lines"] == 0 and info["header_lines"] == 0
        b.append(u, i, v)
        a.finish()
        b.finish()
        assert a.counts() == b.counts()
        for side in (pkg.SIDE_X, pkg.SIDE_Y):
            ca, cb = a.csr(side), b.csr(side)
            assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1]) and np.array_equal(ca[2].view(np.uint32), cb[2].view(np.uint32))
            assert np.array_equal(a.ids(side), b.ids(side))


def test_known_items_reach_the_recommender():
    """generation.getKnownItemIDs() is what recommend() skips (ServerRecommender.java:394-425), and it keeps the entries
    removeSmall pruned from R (IFR:173-211): a user's near-zero entry is not recommended back to him."""
    from oracle import topn_oracle
    rng = np.random.default_rng(5)
    n_users, n_items, k = 60, 500, 8
    lines = []
    for u in range(n_users):
        for i in rng.choice(n_items, 12, replace=False):
            lines.append("%d,%d,%s" % (u, i, "0.00001" if rng.random() < 0.3 else "2"))
    data = ("\n".join(lines) + "\n").encode()
    want = to.expected([data])
    (uid, rp, col, val) = want["csr_x"]
    iid = want["csr_y"][0]
    assert want["known_ptr"][-1] > rp[-1]                          # some known items are not entries of R any more
    X = rng.standard_normal((len(uid), k)).astype(np.float32)
    Y = rng.standard_normal((len(iid), k)).astype(np.float32)
    with ingest.Ingest(0) as g, pkg.ALSCore(k) as core:
        g.set_option(_lib.INGEST_OPT_KNOWN_ITEMS, 1)
        g.append_text(data, True)
        g.finish()
        core.set_factor_rows(pkg.SIDE_X, len(uid))
        core.set_factor_rows(pkg.SIDE_Y, len(iid))
        g.install(core)                                            # hands knownItemIDs over too
        core.set_factors(pkg.SIDE_X, X)
        core.set_factors(pkg.SIDE_Y, Y)
        users = np.arange(len(uid), dtype=np.int64)
        idx, sc, cnt = core.recommend(users, 5)
        for q in range(len(uid)):
            known = want["known_idx"][want["known_ptr"][q]:want["known_ptr"][q + 1]]
            oidx, osc = topn_oracle.recommend(Y, X[q], 5, known)
            assert np.array_equal(idx[q], oidx) and np.array_equal(sc[q].view(np.uint32), osc.view(np.uint32))
        core.set_known_items(None, None)                           # back to the rows of R: the pruned entries come back
        idx2, _, _ = core.recommend(users, 5)
        assert not np.array_equal(idx, idx2)
