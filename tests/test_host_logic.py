"""CPU tests of the product's HOST logic against the oracle (no GPU, no compute kernels):
loader + partition, operator assembly, proximal coefficients, the multifrontal SPD solver's
set-up path and the chordal initialisation.  Also checks that the C-ABI library loads and
exports every symbol declared in include/dpgo_amd.h."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import dpgo_amd
from oracle import g2o as og
from oracle.assemble import assemble_node
from oracle.problem import LOSS_HUBER, LOSS_NONE
from oracle.star import chordal_initialization, GlobalProblem
from oracle.hash import Options as OOptions


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(os.path.dirname(dpgo_amd._HERE), "include", "dpgo_amd.h")).read()
    declared = set(re.findall(r"\b(dpgo_[a-z_A-Z0-9]+)\s*\(", hdr))
    assert declared == set(dpgo_amd.SYMBOLS), declared ^ set(dpgo_amd.SYMBOLS)
    L = dpgo_amd.lib()
    for name in declared:
        assert hasattr(L, name), name


@pytest.mark.parametrize("name,nn", [("tinyGrid3D", 2), ("smallGrid3D", 2), ("sphere2500", 3), ("M3500", 4)])
def test_loader_and_partition(fixtures_dir, name, nn):
    path = os.path.join(fixtures_dir, name + ".g2o")
    G = dpgo_amd.read_g2o(path, nn)
    num_poses, mm = og.read_g2o_file(path)
    assert (G.d, G.num_poses, G.num_edges) == (mm.d, num_poses, len(mm))
    I, J, R, t, kap, tau = G.edges()
    np.testing.assert_array_equal(I, mm.ipose)
    np.testing.assert_array_equal(J, mm.jpose)
    np.testing.assert_allclose(R, mm.R, rtol=0, atol=1e-15)
    np.testing.assert_allclose(t, mm.t, rtol=0, atol=0)
    np.testing.assert_allclose(kap, mm.kappa, rtol=1e-14)
    np.testing.assert_allclose(tau, mm.tau, rtol=1e-14)
    _, meas, g_index = og.partition_measurements(num_poses, mm, nn)
    for a in range(nn):
        info = og.generate_data_info(a, meas[a])
        assert G.node_sizes(a) == (info.n[0], info.n[1], info.m[0], info.m[1])
        nb_node, nb_pose = G.node_neighbours(a)
        want = sorted((b, p) for b, v in info.index.items() if b != a for p in v)
        assert list(zip(nb_node.tolist(), nb_pose.tolist())) == want
        assert G.node_offset(a) == g_index[a][0]


def test_appendix_c_partition_sizes(fixtures_dir):
    """SURVEY.md Appendix C."""
    G = dpgo_amd.read_g2o(os.path.join(fixtures_dir, "smallGrid3D.g2o"), 2)
    assert [G.node_sizes(a) for a in range(2)] == [(63, 25, 135, 30), (62, 25, 132, 30)]
    G = dpgo_amd.read_g2o(os.path.join(fixtures_dir, "M3500.g2o"), 4)
    assert [G.node_sizes(a) for a in range(4)] == [(875, 203, 1259, 258), (875, 211, 1249, 285),
                                                   (875, 255, 1161, 310), (875, 83, 1305, 105)]
    G = dpgo_amd.read_g2o(os.path.join(fixtures_dir, "sphere2500.g2o"), 1)
    assert G.node_sizes(0) == (2500, 0, 4949, 0)


@pytest.mark.parametrize("name,nn,loss", [("smallGrid3D", 2, LOSS_NONE), ("smallGrid3D", 3, LOSS_HUBER),
                                          ("M3500", 4, LOSS_NONE), ("M3500", 4, LOSS_HUBER)])
def test_operator_assembly_matches_oracle(fixtures_dir, name, nn, loss):
    path = os.path.join(fixtures_dir, name + ".g2o")
    G = dpgo_amd.read_g2o(path, nn)
    opt = dpgo_amd.Options.driver(loss)
    num_poses, mm = og.read_g2o_file(path)
    _, meas, _ = og.partition_measurements(num_poses, mm, nn)
    for a in range(nn):
        info = og.generate_data_info(a, meas[a])
        ref = assemble_node(info, mm.d, opt.regularizer, loss == LOSS_NONE)
        names = ["G", "D", "Q"] + (["S", "P", "P0"] if loss == LOSS_NONE else [])
        for nm in names:
            mine = G.node_matrix(a, opt, nm)
            want = getattr(ref, nm)
            diff = abs(mine - want)
            scale = abs(want).max()
            assert diff.max() <= 1e-13 * scale, (nm, a, diff.max(), scale)
        T, N, V = G.node_proximal(a, opt)
        n0, d = info.n[0], mm.d
        np.testing.assert_allclose(T, ref.T, rtol=1e-13)
        Nref = ref.N.toarray()
        Vref = (ref.V if loss != LOSS_NONE else None)
        for i in range(n0):
            np.testing.assert_allclose(N[i], Nref[i, i * d:(i + 1) * d], rtol=1e-12, atol=1e-12)
        if Vref is not None:
            Vd = Vref.toarray()
            for i in range(n0):
                np.testing.assert_allclose(V[i], Vd[i * d:(i + 1) * d, i * d:(i + 1) * d], rtol=1e-12, atol=1e-9)


def test_spd_solver_host_path(fixtures_dir):
    """The multifrontal factorisation (set-up path) solves G_tt and G_RR + lambda I exactly."""
    path = os.path.join(fixtures_dir, "sphere2500.g2o")
    num_poses, mm = og.read_g2o_file(path)
    _, meas, _ = og.partition_measurements(num_poses, mm, 2)
    info = og.generate_data_info(1, meas[1])
    ref = assemble_node(info, 3, 1e-11, True)
    rng = np.random.default_rng(0)
    for A in (ref.Gtt, ref.GRR + 1e-3 * sp.eye(ref.GRR.shape[0])):
        B = rng.standard_normal((A.shape[0], 3))
        X = dpgo_amd.spd_solve_host(A, B)
        r = np.linalg.norm(A @ X - B) / np.linalg.norm(B)
        assert r < 1e-11, r


def test_spd_solver_disconnected_and_tiny():
    A = sp.block_diag([sp.csr_matrix(np.array([[4.0, 1, 0], [1, 3, 1], [0, 1, 5]])), sp.csr_matrix([[2.0]])]).tocsr()
    B = np.arange(8.0).reshape(4, 2)
    X = dpgo_amd.spd_solve_host(A, B, leaf=1)
    np.testing.assert_allclose(A @ X, B, atol=1e-12)


def test_spd_solver_large_components_and_separator_quality():
    """Two large disconnected lattice Laplacians (the situation of a group with several nodes): the components are
    dissected in parallel with minimum-vertex-cover / spectral separators; the solve is exact, and the top separator
    of a 16 x 16 x 12 lattice is close to its smallest cross-section (16 x 12 = 192), not a diagonal level set."""
    def lattice(nx, ny, nz):
        idx = np.arange(nx * ny * nz).reshape(nx, ny, nz)
        e = [(idx[:-1].ravel(), idx[1:].ravel()), (idx[:, :-1].ravel(), idx[:, 1:].ravel()),
             (idx[:, :, :-1].ravel(), idx[:, :, 1:].ravel())]
        i = np.concatenate([a for a, _ in e]); j = np.concatenate([b for _, b in e])
        W = sp.coo_matrix((np.ones(len(i)), (i, j)), shape=(idx.size, idx.size))
        W = W + W.T
        return (sp.diags(np.asarray(W.sum(1)).ravel() + 0.5) - W).tocsr()
    A = sp.block_diag([lattice(16, 16, 12), lattice(15, 14, 13)]).tocsr()
    rng = np.random.default_rng(3)
    B = rng.standard_normal((A.shape[0], 3))
    X = dpgo_amd.spd_solve_host(A, B, leaf=64)
    np.testing.assert_allclose(A @ X, B, atol=1e-9)
    os.environ["DPGO_SPD_COLLAPSE"] = "1"
    try:
        _, _, max_front = dpgo_amd.spd_stats(lattice(16, 16, 12), 64)
    finally:
        del os.environ["DPGO_SPD_COLLAPSE"]
    assert max_front <= 1.35 * 192


@pytest.mark.parametrize("name", ["smallGrid3D", "M3500"])
def test_chordal_initialization_matches_oracle(fixtures_dir, name):
    path = os.path.join(fixtures_dir, name + ".g2o")
    G = dpgo_amd.read_g2o(path, 1)
    X = G.chordal_initialization()
    num_poses, mm = og.read_g2o_file(path)
    Xo = chordal_initialization(num_poses, mm)
    np.testing.assert_allclose(X, Xo, atol=2e-8)
    star = GlobalProblem(num_poses, mm, 1, OOptions.driver())
    assert abs(star.evaluate_f(X) - star.evaluate_f(Xo)) <= 1e-8 * abs(star.evaluate_f(Xo))


def test_cpp_facade_builds_with_plain_gxx_and_reads_partitions(fixtures_dir, tmp_path):
    """include/dpgo_amd.hpp (the reference's C++ names over the C ABI) compiles with g++ alone -- no HIP, no
    torch at the call site -- and its host-only part agrees with the oracle's partition (DPGOProblem n() / m())."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "facade_mm")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "examples", "facade_mm.cpp"), "-o", exe, "-L", os.path.join(root, "dpgo_amd"),
                    "-ldpgo_amd", "-Wl,-rpath," + os.path.join(root, "dpgo_amd")], check=True)
    path = os.path.join(fixtures_dir, "smallGrid3D.g2o")
    out = subprocess.run([exe, "--info", path, "4"], check=True, capture_output=True, text=True).stdout.split("\n")
    num_poses, mm = og.read_g2o_file(path)
    _, meas, g_index = og.partition_measurements(num_poses, mm, 4)
    assert out[0].split() == ["d", "3", "poses", str(num_poses), "edges", str(len(mm)), "nodes", "4"]
    for a in range(4):
        info = og.generate_data_info(a, meas[a])
        assert out[1 + a].split() == ["node", "%d:" % a, "n", str(info.n[0]), str(info.n[1]), "m", str(info.m[0]),
                                      str(info.m[1]), "offset", str(g_index[a][0])]


def test_node_maps_match_oracle_index_sent_recv(fixtures_dir):
    """DPGOProblem::index() / sent() / recv() (DPGOProblem.h:212-225) through dpgo_graph_node_maps."""
    path = os.path.join(fixtures_dir, "torus3D.g2o")
    num_poses, mm = og.read_g2o_file(path)
    _, meas, _ = og.partition_measurements(num_poses, mm, 4)
    G = dpgo_amd.read_g2o(path, 4)
    for a in range(4):
        info = og.generate_data_info(a, meas[a])
        idx = {k: v for k, v in G.node_maps(a, "index")}
        want = {(b, p): tuple(v) for b, poses in info.index.items() for p, v in poses.items()}
        assert idx == want
        for name, ref in (("sent", info.sent), ("recv", info.recv)):
            got = {k: v for k, v in G.node_maps(a, name)}
            want = {(b, p): tuple(v) for b, poses in ref.items() for p, v in poses.items()}
            assert got == want, name


@pytest.mark.parametrize("name", ["smallGrid3D", "M3500"])
def test_g2o_export_round_trips(fixtures_dir, tmp_path, name):
    """dpgo_write_g2o: the exported file reads back to the same measurements (kappa / tau through the loader's own
    formulas, R through quaternion / angle: 1e-12) and carries one VERTEX line per pose with the poses of X."""
    G = dpgo_amd.read_g2o(os.path.join(fixtures_dir, name + ".g2o"), 2)
    X = G.chordal_initialization()
    out = str(tmp_path / "out.g2o")
    assert G.write_g2o(out, X) == 0
    G2 = dpgo_amd.read_g2o(out, 2)
    assert (G2.d, G2.num_poses, G2.num_edges) == (G.d, G.num_poses, G.num_edges)
    a, b = G.edges(), G2.edges()
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    # the fixture's quaternions are not normalised to machine precision (SURVEY: R_e is kept as read): the exported
    # quaternion is that of the nearest rotation
    np.testing.assert_allclose(a[2], b[2], atol=2e-6)
    np.testing.assert_allclose(a[3], b[3], rtol=1e-15)
    np.testing.assert_allclose(a[4], b[4], rtol=1e-12)
    np.testing.assert_allclose(a[5], b[5], rtol=1e-12)
    d, N = G.d, G.num_poses
    verts = [ln.split() for ln in open(out) if ln.startswith("VERTEX")]
    assert len(verts) == N and [int(v[1]) for v in verts] == list(range(N))
    t = np.array([[float(x) for x in v[2:2 + d]] for v in verts])
    np.testing.assert_allclose(t, X[:N], rtol=1e-15)
    if d == 2:
        th = np.array([float(v[4]) for v in verts])
        R = X[N:].reshape(N, 2, 2).transpose(0, 2, 1)       # rows N + d i hold R_i^T
        np.testing.assert_allclose(np.cos(th), R[:, 0, 0], atol=1e-12)
        np.testing.assert_allclose(np.sin(th), R[:, 1, 0], atol=1e-12)


def test_c_abi_rejects_bad_input(fixtures_dir):
    """pose ids outside [0, num_poses), repeated node ids and unknown option values return -1 (no GPU
    needed: the checks run before any device work)."""
    I, J = np.array([0, 1, 5], np.int32), np.array([1, 2, 0], np.int32)
    R, t = np.tile(np.eye(3), (3, 1, 1)), np.zeros((3, 3))
    k = np.ones(3)
    with pytest.raises(ValueError):
        dpgo_amd.graph_from_edges(3, 3, I, J, R, t, k, k, 1)          # pose id 5 >= num_poses
    I[2] = -1
    with pytest.raises(ValueError):
        dpgo_amd.graph_from_edges(3, 3, I, J, R, t, k, k, 1)          # negative pose id
    G = dpgo_amd.read_g2o(os.path.join(fixtures_dir, "tinyGrid3D.g2o"), 2)
    with pytest.raises(RuntimeError):
        dpgo_amd.NodeGroup(G, [0, 0], dpgo_amd.Options.driver())        # repeated node id
    with pytest.raises(RuntimeError):
        dpgo_amd.NodeGroup(G, [0, 7], dpgo_amd.Options.driver())        # node id out of range
    o = dpgo_amd.Options()
    assert o.rescale == dpgo_amd.RESCALE_DYNAMIC and o.max_rescale_count == 5      # DPGO_types.h:128-131
    assert o.preconditioner == dpgo_amd.PRECON_REG_CHOLESKY                        # DPGO_types.h:155
    assert dpgo_amd.Options.driver(LOSS_HUBER).rescale == dpgo_amd.RESCALE_STATIC  # dist_pgo.cpp:105


def test_chordal_initialization_does_not_depend_on_the_thread_count(fixtures_dir, tmp_path):
    """The host PCG applies its operators pose by pose and sums in fixed chunks (chordal.cpp): 1 thread and 4 threads
    give bit-identical initial guesses (the thread count is read once per process: two subprocesses)."""
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import dpgo_amd; "
            "G = dpgo_amd.read_g2o(%r, 2); np.save(sys.argv[1], G.chordal_initialization())"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(fixtures_dir, "torus3D.g2o")))
    outs = []
    for nt in ("1", "4"):
        path = str(tmp_path / ("x%s.npy" % nt))
        subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, DPGO_HOST_THREADS=nt))
        outs.append(np.load(path))
    assert np.array_equal(outs[0], outs[1])


def test_documented_switches_exist_in_the_sources():
    """Every DPGO_* environment switch DESIGN.md names is read somewhere in the library, the driver, bench.py or the
    tools (a table that outlives its switches misleads whoever tunes next; removed switches are named in DESIGN 9 without the
    code quotes this test looks for)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "DESIGN.md")).read()
    names = set(re.findall(r"`(DPGO_[A-Z0-9_]+)", doc))
    assert len(names) > 20
    src = ""
    for pat in ("dpgo_amd/**/*", "tools/**/*", "tests/*.py", "bench.py", "__graft_entry__.py"):
        for f in glob.glob(os.path.join(root, pat), recursive=True):
            if f.endswith((".cpp", ".h", ".hpp", ".hip", ".py", ".sh")):
                src += open(f, errors="ignore").read()
    missing = sorted(n for n in names if n not in src)
    assert not missing, missing
