"""Small and degenerate pose graphs through the HIP path against the oracle: one pose per node, parallel and
reversed edges, a node without intra-node edges, disconnected pieces inside a node, SE(2) and SE(3), trivial and
Huber loss.  (The reference has no tests for these; the oracle restates its arithmetic, including IEEE
inf / NaN where Python would raise.)"""
import os

import numpy as np
import pytest

import dpgo_amd
from oracle import g2o as og
from oracle.hash import Options as OOptions
from oracle.problem import LOSS_HUBER, LOSS_NONE
from oracle.star import DistPGO as ODistPGO, chordal_initialization

pytestmark = pytest.mark.gpu

CASES = {
    # name: (poses, edges (tail, head), nodes)
    "two_poses_parallel_edges": (2, [(0, 1), (0, 1), (1, 0)], 2),
    "ring5_one_pose_per_node": (5, [(0, 1), (1, 2), (2, 3), (3, 4), (4, 0)], 5),
    "ring6_reversed_and_duplicate": (6, [(1, 0), (1, 2), (2, 3), (3, 2), (3, 4), (4, 5), (0, 5), (0, 5)], 3),
    "k4_one_node": (4, [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)], 1),
    "two_rings_joined_across_nodes": (8, [(0, 1), (1, 2), (2, 0), (4, 5), (5, 6), (6, 7), (7, 4), (3, 0), (3, 4), (2, 6)], 2),
}


def _write(path, edges, d, seed):
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        for (i, j) in edges:
            if d == 3:
                q = rng.standard_normal(4)
                q /= np.linalg.norm(q)
                t = rng.standard_normal(3)
                f.write("EDGE_SE3:QUAT %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g "
                        "100 0 0 0 0 0 100 0 0 0 0 100 0 0 0 400 0 0 400 0 400\n" % (i, j, *t, *q))
            else:
                t = rng.standard_normal(2)
                f.write("EDGE_SE2 %d %d %.17g %.17g %.17g 100 0 0 100 0 400\n" % (i, j, *t, rng.standard_normal()))


@pytest.mark.parametrize("d", [3, 2])
@pytest.mark.parametrize("loss", [LOSS_NONE, LOSS_HUBER])
@pytest.mark.parametrize("name", sorted(CASES))
def test_degenerate_graph(tmp_path, name, loss, d):
    _, edges, nn = CASES[name]
    path = str(tmp_path / (name + ".g2o"))
    _write(path, edges, d, seed=len(name) + d)
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    orc = ODistPGO(path, nn, OOptions.driver(loss, True), X0=X0, mm=mm, num_poses=num_poses)
    G = dpgo_amd.read_g2o(path, nn)
    gpu = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(loss, True), X0=X0)
    for it in range(15):
        orc.step(evaluate=False)
        assert gpu.step() == 0
        fo = sum(nd.results.fobj[0] for nd in orc.nodes)
        fg = sum(gpu.group.results(k).fobj for k in range(nn))
        assert np.isfinite(fg)
        np.testing.assert_allclose(fg, fo, rtol=1e-7, atol=1e-9, err_msg="%s it=%d" % (name, it))
    Xg, Xo = gpu.X(), orc.gather()
    if nn == 1:   # one node: G_tt = Laplacian + 1e-11 I is singular along the constant vector (the gauge)
        Xg[:num_poses] -= Xg[:num_poses].mean(axis=0)
        Xo[:num_poses] -= Xo[:num_poses].mean(axis=0)
    np.testing.assert_allclose(Xg, Xo, atol=1e-6)
