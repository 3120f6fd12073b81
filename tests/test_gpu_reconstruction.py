"""the reconstruction metric on the device (mals_reconstruction_error)
against the oracle's restatement of ReconstructionEvaluator.java:91-102."""
import numpy as np
import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [2, 10, 30, 64, 100])
def test_reconstruction_error_matches_oracle(k):
    n_users, n_items = 500, 300
    r_csr, c_csr, Y0 = synth.numpy_problem(n_users, n_items, 20000, k, seed=40 + k)
    with pkg.ALSCore(k) as core:
        core.set_factor_rows(pkg.SIDE_X, n_users)
        core.set_factor_rows(pkg.SIDE_Y, n_items)
        core.set_matrix(pkg.SIDE_X, *r_csr)
        core.set_matrix(pkg.SIDE_Y, *c_csr)
        core.set_factors(pkg.SIDE_Y, Y0)
        errs = []
        for _ in range(3):
            core.half_iteration(pkg.SIDE_X)
            core.half_iteration(pkg.SIDE_Y)
            s, n = core.reconstruction_error()
            X, Y = core.get_factors(pkg.SIDE_X), core.get_factors(pkg.SIDE_Y)
            so, no = oracle.reconstruction_error(*r_csr, X, Y)
            assert n == no == len(r_csr[1])
            assert abs(s - so) <= 1e-11 * max(1.0, abs(so))
            errs.append(s / n)
    assert all(0.0 <= e < 1.0 for e in errs)
    if k >= 10:
        assert errs[-1] <= errs[0] + 1e-9          # ALS keeps improving the reconstruction of the observed entries


def test_empty_and_missing_inputs():
    with pkg.ALSCore(8) as core:
        with pytest.raises(pkg.MalsError):
            core.reconstruction_error()             # no matrix yet
        core.set_factor_rows(pkg.SIDE_X, 3)
        core.set_factor_rows(pkg.SIDE_Y, 2)
        core.set_matrix(pkg.SIDE_X, np.array([0, 0, 0, 0], np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32))
        assert core.reconstruction_error() == (0.0, 0)
