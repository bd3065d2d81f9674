"""The numeric multifrontal factorisation on the GPU (spd_dev.hip: block-column elimination of the fronts with
v_mfma_f64_16x16x4_f64 trailing updates) against scipy on the same matrices: front sizes below, at and above the
32-wide block column and the 128-wide super-block, with ragged last blocks."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import dpgo_amd

pytestmark = pytest.mark.gpu


def _grid_laplacian(shape, rng, diag=1e-3):
    idx = np.arange(int(np.prod(shape))).reshape(shape)
    rows, cols, vals = [], [], []
    for ax in range(len(shape)):
        a = np.take(idx, range(shape[ax] - 1), axis=ax).ravel()
        b = np.take(idx, range(1, shape[ax]), axis=ax).ravel()
        w = rng.uniform(0.5, 2.0, len(a))
        rows += [a, b, a, b]
        cols += [b, a, a, b]
        vals += [-w, -w, w, w]
    n = idx.size
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    return A + diag * sp.eye(n)


@pytest.mark.parametrize("shape,leaf", [((9, 9), 8), ((37, 41), 16), ((70, 70), 32), ((13, 14, 15), 32), ((21, 22, 23), 64),
                                        ((30, 31, 29), 128)])
def test_device_factor_solves_like_scipy(shape, leaf):
    rng = np.random.default_rng(sum(shape))
    A = _grid_laplacian(shape, rng)
    n = A.shape[0]
    B = rng.standard_normal((n, 3))
    nnz, levels, max_front = dpgo_amd.spd_stats(A, leaf)
    X = dpgo_amd.spd_solve_host(A, B, leaf=leaf)            # factorisation on the GPU, sweeps on the host
    ref = spla.splu(A.tocsc()).solve(B)
    np.testing.assert_allclose(X, ref, rtol=0, atol=1e-9 * np.abs(ref).max())
    assert np.abs(A @ X - B).max() <= 1e-10 * max(1.0, np.abs(B).max()) * n
    if len(shape) == 3 and min(shape) >= 21:
        assert max_front > 256                              # the wide (K = 128) pass was exercised
