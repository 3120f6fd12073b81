"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's top-N scoring (SURVEY.md section 8(f)
row 4).  Not product code: only tests/ and tools/bench_topn.py's cpu_baseline may import it.

  RecommendIterator.next   online/src/net/myrrix/online/RecommendIterator.java:62-109
      skip known items; score = (float) (sum over the query's vectors of dot / count)
  SimpleVectorMath.dot     common/src/net/myrrix/common/math/SimpleVectorMath.java:34-41 (fp32 products, fp64 sum)
  TopN.selectTopN          common/src/net/myrrix/common/TopN.java:49-67,117-128,136-145 (bounded priority queue,
                           result sorted by value descending)
Pinned by TopNTest (common/test/net/myrrix/common/TopNTest.java:30-77) in tests/test_topn_oracle.py.
Among equal scores the reference's order is the hash-slot order of its maps (unspecified); this
restatement, like the device path, orders ties by ascending item index.
"""
import heapq

import numpy as np


def _seq_sum(p):
    """double dot = 0.0; dot += p[0]; dot += p[1]; ... (SVM:36-39): starts from +0.0, so a sum of negative zeros is +0.0"""
    z = np.zeros((len(p), 1), np.float64)
    return np.cumsum(np.concatenate([z, p.astype(np.float64)], axis=1), axis=1)[:, -1]


def scores(Y, x):
    """(float) dot(Y_i, x) for every item: fp32 products, sequential fp64 sum, cast to fp32."""
    p = (np.asarray(Y, np.float32) * np.asarray(x, np.float32)[None, :]).astype(np.float32)
    d = _seq_sum(p)
    return d.astype(np.float32)


def select_top_n(items, n):
    """TopN.selectTopN over an iterator of (item, value): the queue may hold n+1 entries while streaming
    (TopN.java:56: size() > n), the extra one is dropped at the end (:120-122)."""
    heap = []                                             # min-heap on value, like ByValueAscComparator
    for item, value in items:
        if len(heap) > n:
            if value > heap[0][0]:
                heapq.heapreplace(heap, (value, -item, item))
        else:
            heapq.heappush(heap, (value, -item, item))
    while len(heap) > n:
        heapq.heappop(heap)
    return [(it, v) for v, _, it in sorted(heap, key=lambda t: (-t[0], t[2]))]


def scores_to_many(Y, vectors):
    """RecommendIterator.java:93-104 for a query of several vectors: (float)((dot_1 + ... + dot_n) / n), the dots
    added in fp64 in the order of the vectors."""
    vectors = np.asarray(vectors, np.float32)
    total = np.zeros(len(Y), np.float64)
    for x in vectors:
        p = (np.asarray(Y, np.float32) * x[None, :]).astype(np.float32)
        total = total + _seq_sum(p)
    return (total / float(len(vectors))).astype(np.float32)


def recommend(Y, x, how_many, known=None, tags=None):
    """Returns (item indices, scores), best first, ties by ascending index.  x: one vector, or an array of several
    (recommendToMany).  tags: userTagIDs as item indices -- RecommendIterator.java:72 returns null for them before it
    looks at anything else."""
    x = np.asarray(x, np.float32)
    s = scores(Y, x) if x.ndim == 1 else scores_to_many(Y, x)
    ok = np.ones(len(s), bool)
    if tags is not None and len(tags):
        ok[np.asarray(tags, np.int64)] = False            # RecommendIterator.java:72
    if known is not None and len(known):
        ok[np.asarray(known, np.int64)] = False           # RecommendIterator.java:75-82
    idx = np.flatnonzero(ok)
    order = np.lexsort((idx, -s[idx].astype(np.float64)))[:how_many]
    return idx[order], s[idx][order]
