"""Pins the top-N oracle to the reference's TopNTest (common/test/net/myrrix/common/TopNTest.java:30-77)
and checks its two selection routines against each other."""
import numpy as np

from oracle import topn_oracle as to


def candidates(n):                                                   # makeNCandidates: item i has value i
    return [(i, np.float32(i)) for i in range(1, n + 1)]


def test_top_n_known_answers():
    assert to.select_top_n(iter([]), 2) == []                        # testEmpty
    top3 = to.select_top_n(iter(candidates(3)), 3)                   # testTopExactly
    assert [t[0] for t in top3] == [3, 2, 1] and top3[0][1] == 3.0 and top3[2][1] == 1.0
    top3 = to.select_top_n(iter(candidates(4)), 3)                   # testTopPlusOne
    assert [t[0] for t in top3] == [4, 3, 2]
    top3 = to.select_top_n(iter(candidates(20)), 3)                  # testTopOfMany
    assert [t[0] for t in top3] == [20, 19, 18] and top3[2][1] == 18.0


def test_recommend_equals_streaming_top_n():
    rng = np.random.default_rng(1)
    Y = rng.standard_normal((500, 12)).astype(np.float32)
    x = rng.standard_normal(12).astype(np.float32)
    known = [3, 17, 256]
    idx, sc = to.recommend(Y, x, 10, known)
    s = to.scores(Y, x)
    stream = ((i, s[i]) for i in range(len(s)) if i not in known)
    assert [(int(i), float(v)) for i, v in zip(idx, sc)] == [(i, float(v)) for i, v in to.select_top_n(stream, 10)]
    # fp32 products, fp64 sum
    assert sc[0] == np.float32(sum(float(np.float32(a * b)) for a, b in zip(Y[idx[0]], x)))
