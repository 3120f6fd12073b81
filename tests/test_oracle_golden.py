"""Pins the CPU oracle (oracle/als_oracle.c) to every known-answer vector the reference's own
tests hold for the ALS hot path."""
import json
import os

import numpy as np
import pytest

from oracle import oracle

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                     "reference_known_answers.json")))


def run_case(case, threads=1):
    R = np.array(case["R"], dtype=np.float32)
    r_csr, c_csr = oracle.dense_to_csr(R)
    Y0 = np.array(case["Y0"], dtype=np.float32)
    X, Y, iters, conv = oracle.als_call(
        r_csr, c_csr, R.shape[0], R.shape[1], Y0, case["features"], alpha=1.0, lam=0.1,
        flags=case["flags"], conv_threshold=case["threshold"],
        max_iterations=case["max_iterations"], random_y=False, threads=threads)
    return X, Y, iters, conv


@pytest.mark.parametrize("name,iters_expected", [("als_default", 28), ("als_reconstruct_r", 34),
                                                 ("als_negative_input", 19)])
def test_als_known_answers(name, iters_expected):
    case = GOLDEN[name]
    X, Y, iters, conv = run_case(case)
    P = oracle.multiply_xyt(X, Y).astype(np.float32)  # the test casts getRow() to float
    expected = np.array(case["expected_XYT"], dtype=np.float32)
    assert P.shape == expected.shape
    assert np.max(np.abs(P - expected)) <= case["tol"] + 1e-9, (P, expected)
    # derived (not from the reference): iteration count at which the restatement converges
    assert iters == iters_expected
    assert conv < case["threshold"]


def test_als_known_answers_multithreaded_same_result():
    case = GOLDEN["als_default"]
    X1, Y1, _, _ = run_case(case, threads=1)
    X4, Y4, _, _ = run_case(case, threads=4)
    assert np.array_equal(X1, X4) and np.array_equal(Y1, Y4)


def test_gramian_known_answer():
    g = GOLDEN["gramian"]
    G = oracle.gramian(np.array(g["M"], dtype=np.float32))
    assert np.max(np.abs(G - np.array(g["expected_MTM"]))) <= g["tol"]


def test_vector_math_known_answers():
    v = GOLDEN["vector_math"]
    assert abs(oracle.dot(v["v1"], v["v2"]) - v["dot"]) <= v["tol"]
    assert abs(oracle.norm(v["v1"]) - v["norm1"]) <= v["tol"]
    assert abs(oracle.norm(v["v2"]) - v["norm2"]) <= v["tol"]


def test_rrqr_matches_generic_solve_and_flags_singular():
    rng = np.random.default_rng(1234567890)
    for k in (1, 2, 7, 30, 64):
        A = rng.standard_normal((k + 5, k))
        W = A.T @ A + 0.1 * np.eye(k)
        b = rng.standard_normal(k)
        x = oracle.rrqr_solve(W, b)
        ref = np.linalg.solve(W, b)
        assert np.allclose(x, ref.astype(np.float32), rtol=2e-6, atol=1e-7)
    # rank-2 4x4 matrix: singular at threshold 1e-5, apparent rank 2 (CMLSS:46-54 semantics)
    B = rng.standard_normal((4, 2))
    with pytest.raises(oracle.SingularMatrix) as ei:
        oracle.rrqr_solve(B @ B.T, np.ones(4))
    assert ei.value.apparent_rank == 2


def test_empty_row_gives_zero_vector():
    # an empty row has W = G, b = 0 => x = 0 when G is non-singular.
    rng = np.random.default_rng(7)
    M = rng.standard_normal((20, 3)).astype(np.float32)
    row_ptr = np.array([0, 2, 2, 3], dtype=np.int64)
    col = np.array([1, 5, 7], dtype=np.int32)
    val = np.array([1.0, 2.0, -1.0], dtype=np.float32)
    out = oracle.half_iteration(row_ptr, col, val, M)
    assert np.all(out[1] == 0.0)
    # negative-only row: contributes to W, nothing to b => x = 0 as well
    assert np.all(out[2] == 0.0)
    assert np.any(out[0] != 0.0)
