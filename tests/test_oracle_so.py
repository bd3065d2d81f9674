"""oracle.problem.project_to_SOdn against the REFERENCE's own AVX2 kernels.

tests/golden/so_ref.npz holds seeded inputs and the outputs of DPGO::internal::project_to_SO3 / project_to_SO2
(C++/DPGO/src/internal/project_to_SOd.cpp:7-33, 97-196: McAdams-style Jacobi SVD, 8 sweeps), produced by
oracle/_ref/so_ref (oracle/ref_so3/Makefile builds it from the reference's sources where they lie;
tests/golden/make_so_golden.py wrote the file).  This pins row a9 of the path: the oracle's nearest-rotation map
IS the reference's, to the accuracy the reference's fixed sweep count reaches."""
import os

import numpy as np

from oracle.problem import project_to_SOdn


def _load(golden_dir):
    z = np.load(os.path.join(golden_dir, "so_ref.npz"))
    return z["A3"], z["U3"], z["A2"], z["U2"]


def well_conditioned(A):
    """The projection is unique and smooth where sigma_2 + sign(det) sigma_3 is not small."""
    s = np.linalg.svd(A, compute_uv=False)
    return (s[:, 1] + s[:, 2] * np.sign(np.linalg.det(A))) > 1e-3 * s[:, 0]


def test_so3_matches_reference_kernel(golden_dir):
    A, U, _, _ = _load(golden_dir)
    n = len(A)
    got = project_to_SOdn(A.reshape(3 * n, 3), 3).reshape(n, 3, 3)
    # the reference's outputs are rotations to machine precision
    np.testing.assert_allclose(np.einsum("nij,nkj->nik", U, U), np.broadcast_to(np.eye(3), U.shape), atol=1e-14)
    np.testing.assert_allclose(np.linalg.det(U), 1.0, atol=1e-14)
    well = well_conditioned(A)
    assert well.sum() >= 1190
    np.testing.assert_allclose(got[well], U[well], rtol=0, atol=1e-9)      # measured: 2.3e-10 (8 Jacobi sweeps)
    # everywhere (incl. rank-deficient inputs, where the projection is not unique): same distance to the input
    d_got = np.linalg.norm((got - A).reshape(n, -1), axis=1)
    d_ref = np.linalg.norm((U - A).reshape(n, -1), axis=1)
    np.testing.assert_allclose(d_got, d_ref, rtol=1e-8, atol=1e-12)


def test_so2_matches_reference_kernel(golden_dir):
    _, _, A, U = _load(golden_dir)
    n = len(A)
    got = project_to_SOdn(A.reshape(2 * n, 2), 2).reshape(n, 2, 2)
    np.testing.assert_allclose(got, U, rtol=0, atol=1e-15)
    np.testing.assert_array_equal(U[50], np.eye(2))      # c^2 + s^2 < 1e-32: the guard of traits.cpp:10 gives the identity
    np.testing.assert_array_equal(U[51], np.eye(2))
