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


@pytest.mark.parametrize("blocks", ["1", "3", "8"])
def test_roots_as_one_triangle_solve_like_the_full_product(tmp_path, blocks):
    """The fused roots of the elimination trees stored as the lower triangle of L11^-T L11^-1 in 64 x 64 blocks
    (k_root_sym + k_root_combine: every block serves its row block and, transposed, its column block; partial sums are
    combined in a fixed order) against the full product (k_spd_level MODE 2) and against the operator itself: both SPD
    solves of a node (L_.solve and reg_Chol_precon_.solve, DPGOProblem.h:291, DPGOProblem.cpp:568, 592) on a lattice whose
    roots span several blocks with ragged last ones, 1, 3 and 8 blocks per item, run twice (bit-identical)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import dpgo_amd
from dpgo_amd import synthetic
g = synthetic.grid(18, 17, 12, 14000, seed=11)
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 2)
grp = dpgo_amd.NodeGroup(G, [0, 1], dpgo_amd.Options.driver(1, True))
out = {}
for k in range(2):
    n0 = G.node_sizes(k)[0]
    rng = np.random.default_rng(5 + k)
    B = np.zeros((4 * n0, 3))
    B[:] = rng.standard_normal(B.shape)
    for rep in range(2):
        out["tt%%d_%%d" %% (k, rep)] = grp.debug_apply(k, "solve_tt", B, 4 * n0)[:n0]
        out["rr%%d_%%d" %% (k, rep)] = grp.debug_apply(k, "solve_rr", B, 4 * n0)[n0:]
    Z = np.zeros((4 * n0, 3)); Z[:n0] = out["tt%%d_0" %% k]
    out["res%%d" %% k] = grp.debug_apply(k, "G", Z, 4 * n0)[:n0] - B[:n0]
    out["B%%d" %% k] = B
out["stats"] = np.array([grp.solver_stats()["nnz_tt"], grp.solver_stats()["nnz_rr"]])
np.savez(sys.argv[1], **out)
""" % root

    def run(tag, **env):
        path = str(tmp_path / (tag + ".npz"))
        subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, **env))
        return np.load(path)

    full = run("full", DPGO_SPD_ROOT_SYM="0")
    tri = run("tri", DPGO_SPD_ROOT_SYM="1", DPGO_SPD_ROOT_SYM_BLOCKS=blocks, DPGO_SPD_DUMP="1")
    for k in range(2):
        for what in ("tt", "rr"):
            a, b = tri["%s%d_0" % (what, k)], full["%s%d_0" % (what, k)]
            assert np.array_equal(a, tri["%s%d_1" % (what, k)])                # the same bits on a second solve
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-11 * np.abs(b).max(), err_msg="%s node %d" % (what, k))
        assert np.abs(tri["res%d" % k]).max() <= 1e-9 * np.abs(tri["B%d" % k]).max()
