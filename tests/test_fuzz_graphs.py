"""Randomised differential test: small random pose graphs (ring + random chords, parallel edges allowed), random
number of nodes, SE(2) / SE(3), all four losses, MM and AMM -- the HIP path against the oracle after every outer
iteration.  Seeds are fixed, so the cases are the same on every run.  Consistent noisy measurements around a random
ground truth keep the problems well posed (the reference's regime); outliers exercise the robust losses."""
import os

import numpy as np
import pytest

import dpgo_amd
from oracle import g2o as og
from oracle.hash import Options as OOptions
from oracle.star import DistPGO as ODistPGO, chordal_initialization

pytestmark = pytest.mark.gpu


def _rot(d, rng, scale=1.0):
    if d == 2:
        th = scale * rng.standard_normal()
        return np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    w = scale * rng.standard_normal(3)
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _quat(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
    y = np.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2
    z = np.sqrt(max(0.0, 1 - R[0, 0] - R[1, 1] + R[2, 2])) / 2
    x = np.copysign(x, R[2, 1] - R[1, 2])
    y = np.copysign(y, R[0, 2] - R[2, 0])
    z = np.copysign(z, R[1, 0] - R[0, 1])
    return x, y, z, w


def _write_case(path, seed):
    rng = np.random.default_rng(seed)
    d = int(rng.choice([2, 3]))
    n = int(rng.integers(6, 60))
    nn = int(rng.integers(1, min(6, n // 2) + 1))
    Rg = [_rot(d, rng) for _ in range(n)]
    tg = [3.0 * rng.standard_normal(d) for _ in range(n)]
    edges = [(i, (i + 1) % n) for i in range(n)]
    for _ in range(int(rng.integers(n // 2, 2 * n))):
        i, j = rng.integers(0, n, 2)
        if i != j:
            edges.append((int(i), int(j)))
    with open(path, "w") as f:
        for (i, j) in edges:
            outlier = rng.random() < 0.1
            Rij = Rg[i].T @ Rg[j] @ _rot(d, rng, 0.05)
            tij = Rg[i].T @ (tg[j] - tg[i]) + 0.05 * rng.standard_normal(d)
            if outlier:
                Rij, tij = _rot(d, rng), 3.0 * rng.standard_normal(d)
            if d == 3:
                f.write("EDGE_SE3:QUAT %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g "
                        "100 0 0 0 0 0 100 0 0 0 0 100 0 0 0 400 0 0 400 0 400\n" % (i, j, *tij, *_quat(Rij)))
            else:
                f.write("EDGE_SE2 %d %d %.17g %.17g %.17g 100 0 0 100 0 400\n"
                        % (i, j, *tij, np.arctan2(Rij[1, 0], Rij[0, 0])))
    return d, n, nn, int(rng.integers(0, 4)), bool(rng.integers(0, 2))


@pytest.mark.parametrize("seed", range(40))
def test_random_graph(tmp_path, seed):
    path = str(tmp_path / ("fuzz%d.g2o" % seed))
    d, n, nn, loss, acc = _write_case(path, 1000 + seed)
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    orc = ODistPGO(path, nn, OOptions.driver(loss, acc), X0=X0, mm=mm, num_poses=num_poses)
    G = dpgo_amd.read_g2o(path, nn)
    gpu = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(loss, acc), X0=X0)
    tag = "seed=%d d=%d n=%d nodes=%d loss=%d acc=%s" % (seed, d, n, nn, loss, acc)
    for it in range(12):
        orc.step(evaluate=False)
        assert gpu.step() == 0, tag
        fo = sum(nd.results.fobj[0] for nd in orc.nodes)
        fg = sum(gpu.group.results(k).fobj for k in range(nn))
        np.testing.assert_allclose(fg, fo, rtol=1e-7, atol=1e-9, err_msg="%s it=%d" % (tag, it))
    Xg, Xo = gpu.X(), orc.gather()
    if nn == 1:
        Xg[:num_poses] -= Xg[:num_poses].mean(axis=0)
        Xo[:num_poses] -= Xo[:num_poses].mean(axis=0)
    np.testing.assert_allclose(Xg, Xo, atol=1e-6, err_msg=tag)


@pytest.mark.parametrize("seed", range(12))
def test_random_graph_star(tmp_path, seed):
    """The same random graphs through AMM-PGO* (master with the global objective): F, fobj and the branches taken."""
    from oracle.star import DPGOStar as ODPGOStar
    path = str(tmp_path / ("fuzzstar%d.g2o" % seed))
    d, n, nn, loss, _ = _write_case(path, 5000 + seed)
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    orc = ODPGOStar(path, nn, OOptions.driver(loss, True), mm=mm, num_poses=num_poses)
    orc.initialize(X0)
    G = dpgo_amd.read_g2o(path, nn)
    gpu = dpgo_amd.DPGOStar(G, dpgo_amd.Options.driver(loss, True))
    assert gpu.initialize(X0) == 0
    tag = "seed=%d d=%d n=%d nodes=%d loss=%d" % (seed, d, n, nn, loss)
    for it in range(10):
        orc.step()
        assert gpu.step() == 0, tag
        s = gpu.state()
        np.testing.assert_allclose(s["F"], orc.F, rtol=1e-7, atol=1e-9, err_msg="%s it=%d F" % (tag, it))
        np.testing.assert_allclose(s["fobj"], orc.fobj, rtol=1e-7, atol=1e-9, err_msg="%s it=%d fobj" % (tag, it))


@pytest.mark.parametrize("seed", range(16))
def test_random_graph_dynamic_rescale(tmp_path, seed):
    """The random graphs with Rescale::Dynamic (DPGO::Options' default) and a robust loss: nodes are rescaled at
    different iterations, so most updates re-linearise a subset of the group; per-node fobj / Gk after every iteration."""
    path = str(tmp_path / ("fuzzdyn%d.g2o" % seed))
    d, n, nn, loss, acc = _write_case(path, 9000 + seed)
    loss = 1 + loss % 3                       # Huber, GM, Welsch
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    oo = OOptions.driver(loss, acc)
    oo.rescale = 1
    orc = ODistPGO(path, nn, oo, X0=X0, mm=mm, num_poses=num_poses)
    G = dpgo_amd.read_g2o(path, nn)
    gpu = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(loss, acc, rescale=1), X0=X0)
    tag = "seed=%d d=%d n=%d nodes=%d loss=%d acc=%s" % (seed, d, n, nn, loss, acc)
    for it in range(20):
        orc.step(evaluate=False)
        assert gpu.step() == 0, tag
        for a in range(nn):
            ro, rg = orc.nodes[a].results, gpu.group.results(a)
            np.testing.assert_allclose(rg.fobj, ro.fobj[0], rtol=1e-7, atol=1e-9, err_msg="%s it=%d node=%d fobj" % (tag, it, a))
            np.testing.assert_allclose(rg.Gk, ro.Gk, rtol=1e-7, atol=1e-9, err_msg="%s it=%d node=%d Gk" % (tag, it, a))
    Xg, Xo = gpu.X(), orc.gather()
    if nn == 1:
        Xg[:num_poses] -= Xg[:num_poses].mean(axis=0)
        Xo[:num_poses] -= Xo[:num_poses].mean(axis=0)
    np.testing.assert_allclose(Xg, Xo, atol=1e-6, err_msg=tag)
