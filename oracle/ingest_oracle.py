"""TEST INFRASTRUCTURE ONLY -- CPU restatement of what the reference does to the parsed records of its
input files.  Not product code: only tests/ may import it.

Follows, record by record and in file order:
  InputFilesReader.readInputFiles  online-local/src/net/myrrix/online/generation/InputFilesReader.java:165-171
      NaN value -> MatrixUtils.remove, otherwise MatrixUtils.addTo
  MatrixUtils.addTo / addToByRow   common/src/net/myrrix/common/math/MatrixUtils.java:64-92
  MatrixUtils.remove / removeByRow common/.../MatrixUtils.java:102-125   (an emptied row is deleted)
  FastByIDFloatMap.increment       common/.../collection/FastByIDFloatMap.java:129-138 (fp32 add)
  InputFilesReader.removeSmall     IFR:200-211   (|v| < zeroThreshold removed; rows are kept)
Pinned by the reference's MatrixUtilsTest.testAddTo / testRemove (MatrixUtilsTest.java:33-60) in
tests/test_ingest_oracle.py; removeSmall has no reference test (restated from the source, unpinned).
"""
import numpy as np


def read_input_records(user_ids, item_ids, values, zero_threshold=1.0e-4):
    """Returns (RbyRow, RbyColumn): dict id -> dict id -> np.float32, like the reference's maps."""
    by_row, by_col = {}, {}
    for u, i, v in zip(user_ids.tolist(), item_ids.tolist(), np.asarray(values, np.float32)):
        if np.isnan(v):
            for a, b, M in ((u, i, by_row), (i, u, by_col)):          # MU:102-125
                row = M.get(a)
                if row is not None:
                    row.pop(b, None)
                    if not row:
                        del M[a]
        else:
            for a, b, M in ((u, i, by_row), (i, u, by_col)):          # MU:64-92, FBIFM:129-138
                row = M.setdefault(a, {})
                cur = row.get(b)
                row[b] = v if cur is None else np.float32(cur + v)
    thr = np.float32(zero_threshold)
    for M in (by_row, by_col):                                         # IFR:200-211
        for row in M.values():
            for b in [b for b, v in row.items() if abs(v) < thr]:
                del row[b]
    return by_row, by_col


def to_csr(M, col_index):
    """dict-of-dicts -> (ascending row ids, row_ptr, col, val) with columns ascending within a row."""
    ids = np.array(sorted(M.keys()), dtype=np.int64)
    rp = [0]
    col, val = [], []
    for a in ids.tolist():
        ent = sorted((col_index[b], v) for b, v in M[a].items())
        col.extend(c for c, _ in ent)
        val.extend(v for _, v in ent)
        rp.append(len(col))
    return ids, np.array(rp, dtype=np.int64), np.array(col, dtype=np.int32), np.array(val, dtype=np.float32)


def expected_matrices(user_ids, item_ids, values, zero_threshold=1.0e-4):
    by_row, by_col = read_input_records(user_ids, item_ids, values, zero_threshold)
    u_index = {u: k for k, u in enumerate(sorted(by_row.keys()))}
    i_index = {i: k for k, i in enumerate(sorted(by_col.keys()))}
    return to_csr(by_row, i_index), to_csr(by_col, u_index)
