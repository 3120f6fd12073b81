"""Pins the ingest oracle (oracle/ingest_oracle.py) to the reference's own known-answer tests of the
two primitives it is built from: MatrixUtilsTest.testAddTo / testRemove (MatrixUtilsTest.java:33-60)."""
import numpy as np

from oracle import ingest_oracle as io

NaN = float("nan")


def recs(*r):
    a = np.array(r, dtype=np.float64).reshape(-1, 3)
    return a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2].astype(np.float32)


def test_matrix_utils_add_to():                                     # MatrixUtilsTest.java:34-49
    by_row, by_col = io.read_input_records(*recs((0, 0, -1.0), (4, 1, 2.0)))
    assert by_row[0][0] == np.float32(-1.0) and by_col[0][0] == np.float32(-1.0)
    assert 1 not in by_row
    assert by_row[4][1] == np.float32(2.0) and by_col[1][4] == np.float32(2.0)
    assert 0 not in by_row[4]                                       # assertNaN(byRow.get(4).get(0))


def test_matrix_utils_remove():                                     # MatrixUtilsTest.java:51-60
    by_row, by_col = io.read_input_records(*recs((0, 0, -1.0), (4, 1, 2.0), (0, 0, NaN)))
    assert 0 not in by_row                                          # an emptied row is deleted
    assert by_row[4][1] == np.float32(2.0) and by_col[1][4] == np.float32(2.0)


def test_stream_order_semantics():
    # add, add (fp32 sum), remove, add again: the entry restarts from the last value (FBIFM:129-138)
    by_row, _ = io.read_input_records(*recs((7, 3, 1.5), (7, 3, 2.25), (7, 3, NaN), (7, 3, 0.5), (7, 3, 0.25)))
    assert by_row[7][3] == np.float32(0.75)
    # removing something that is not there is a no-op; a later add creates it
    by_row, by_col = io.read_input_records(*recs((1, 1, NaN), (1, 1, 3.0)))
    assert by_row[1][1] == np.float32(3.0) and by_col[1][1] == np.float32(3.0)
    # fp32 accumulation in record order, not a double sum
    vals = [16777216.0, 1.0, 1.0]
    by_row, _ = io.read_input_records(*recs(*[(2, 2, v) for v in vals]))
    assert by_row[2][2] == np.float32(16777216.0)


def test_remove_small_keeps_the_row():                              # IFR:200-211
    by_row, by_col = io.read_input_records(*recs((5, 9, 1.0), (5, 9, -1.0), (6, 9, 2.0), (5, 8, 0.00005)))
    assert by_row[5] == {} and 5 in by_row                          # both entries pruned, the row stays
    assert by_col[9] == {6: np.float32(2.0)} and by_col[8] == {}
    (uid, rp, col, val), (iid, cp, ccol, cval) = io.expected_matrices(*recs((5, 9, 1.0), (5, 9, -1.0), (6, 9, 2.0), (5, 8, 0.00005)))
    assert uid.tolist() == [5, 6] and rp.tolist() == [0, 0, 1] and col.tolist() == [1] and val.tolist() == [2.0]
    assert iid.tolist() == [8, 9] and cp.tolist() == [0, 0, 1] and ccol.tolist() == [1]
