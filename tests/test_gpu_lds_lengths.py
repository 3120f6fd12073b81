"""Row lengths across every boundary of the LDS-staged gather at k = 128 (csrc/lds_kernels.h): 32-entry super-steps, 64-entry
chunks (three chunk buffers rotating per row and across rows), the ring refilled with the NEXT row's first super-step during
a row's last one.  Every length 0 .. 200 and the neighbourhoods of the larger powers of two, shuffled so that short and long
rows follow each other in every order a wave's list can hold, through the fused rows kernel (MODE 0) and, with a small
segment size, through the segments kernel (MODE 1) + finish kernels; against the oracle (ALS:432-504) and against the
register-staged kernels of round 3 (MALS_LDS_GATHER=0)."""
import os

import numpy as np
import pytest

import myrrix_recommender_amd as pkg  # noqa: F401
from myrrix_recommender_amd import _lib
from oracle import oracle
from tests.test_gpu_dual import rel, rows_problem, solve_x

pytestmark = pytest.mark.gpu
K = 128


def boundary_lengths():
    around = [n + d for n in (256, 512, 1024, 2048, 4096) for d in (-33, -32, -31, -1, 0, 1, 31, 32, 33)]
    return np.concatenate([np.arange(0, 201), np.array(around)])


@pytest.mark.parametrize("segment_nnz", [0, 192])
@pytest.mark.parametrize("order_seed", [1, 2])
def test_every_boundary_length_through_the_lds_kernels(segment_nnz, order_seed):
    lengths = boundary_lengths()
    np.random.default_rng(order_seed).shuffle(lengths)
    csr, M = rows_problem(lengths, 6000, K, seed=40 + order_seed)
    kw = dict(solve_mode=_lib.SOLVE_DIRECT)
    if segment_nnz:
        kw["segment_nnz"] = segment_nnz
    X, st = solve_x(K, csr, M, **kw)
    assert st["rows_dual"] == 0 and st["rows_solved"] == len(lengths)
    Xo = oracle.half_iteration(*csr, M, threads=4)
    per_row = np.linalg.norm(X - Xo, axis=1) / np.maximum(np.linalg.norm(Xo, axis=1), 1e-30)
    assert rel(X, Xo) < 1e-5 and per_row.max() < 1e-4, (rel(X, Xo), int(lengths[per_row.argmax()]), per_row.max())
    assert np.all(X[lengths == 0] == 0.0)
    old = os.environ.get("MALS_LDS_GATHER")
    os.environ["MALS_LDS_GATHER"] = "0"
    try:
        Xr, _ = solve_x(K, csr, M, **kw)
    finally:
        if old is None:
            del os.environ["MALS_LDS_GATHER"]
        else:
            os.environ["MALS_LDS_GATHER"] = old
    assert rel(X, Xr) < 2e-6, rel(X, Xr)


def test_mode_flags_through_the_lds_kernels():
    """reconstructR and lossIgnoresUnspecified (ALS:466-469, 524-539: another weight rule, an image of zeros under W) on the
    same boundary lengths -- without the Gramian under it a row of fewer entries than features is singular for the
    reference as well, so those modes get the lengths from 160 up."""
    for flags in (1, 2, 3):
        lengths = boundary_lengths()
        if flags & 2:
            lengths = lengths[lengths >= 160]
        np.random.default_rng(3).shuffle(lengths)
        csr, M = rows_problem(lengths, 6000, K, seed=43)
        X, _ = solve_x(K, csr, M, flags=flags, solve_mode=_lib.SOLVE_DIRECT)
        Xo = oracle.half_iteration(*csr, M, flags=flags, threads=4)
        assert rel(X, Xo) < 1e-4, (flags, rel(X, Xo))
