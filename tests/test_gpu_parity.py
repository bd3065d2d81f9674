"""GPU parity: the HIP path (through the C ABI) against the oracle on the same inputs."""
import os

import numpy as np
import pytest

import dpgo_amd
from oracle import g2o as og
from oracle.hash import Options as OOptions
from oracle.problem import LOSS_HUBER, LOSS_NONE, LOSS_WELSCH, LOSS_GM, project_to_SOdn
from oracle.star import DistPGO as ODistPGO, chordal_initialization

pytestmark = pytest.mark.gpu


def _oracle_opts(loss, acc, **kw):
    o = OOptions.driver(loss, acc)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _pair(fixtures_dir, name, nn, loss, acc, **kw):
    path = os.path.join(fixtures_dir, name + ".g2o")
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    orc = ODistPGO(path, nn, _oracle_opts(loss, acc, **kw), X0=X0, mm=mm, num_poses=num_poses)
    G = dpgo_amd.read_g2o(path, nn)
    gpu = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(loss, acc, **kw), X0=X0)
    return orc, gpu


def test_projection_kernel(fixtures_dir):
    """k_rot_op(mode 2) == nearest rotation (polar factor with det fix), incl. reflections and
    near-singular inputs.  Tolerance 1e-12 absolute on orthonormal outputs."""
    G = dpgo_amd.read_g2o(os.path.join(fixtures_dir, "sphere2500.g2o"), 1)
    grp = dpgo_amd.NodeGroup(G, [0], dpgo_amd.Options.driver())
    n0 = 2500
    rng = np.random.default_rng(1)
    A = rng.standard_normal((n0, 3, 3))
    A[:200] *= 1e-3
    A[200:400] *= 1e3
    q, _ = np.linalg.qr(rng.standard_normal((400, 3, 3)))
    A[400:800] = q + 1e-6 * rng.standard_normal((400, 3, 3))       # near rotations / reflections
    A[800:1000] = q[:200] * np.array([3.0, 2.0, 1e-9])   # nearly rank 2
    A[1000] = 2 * np.eye(3)
    A[1001] = np.diag([1.0, 1.0, -1.0])
    out = grp.debug_apply(0, "project", A.reshape(3 * n0, 3), 3 * n0).reshape(n0, 3, 3)
    ref = project_to_SOdn(A.reshape(3 * n0, 3), 3).reshape(n0, 3, 3)
    np.testing.assert_allclose(np.einsum("nij,nkj->nik", out, out), np.broadcast_to(np.eye(3), (n0, 3, 3)), atol=1e-13)
    np.testing.assert_allclose(np.linalg.det(out), 1.0, atol=1e-13)
    # where the projection is well conditioned the two must agree tightly
    s = np.linalg.svd(A, compute_uv=False)
    well = (s[:, 1] + s[:, 2] * np.sign(np.linalg.det(A))) > 1e-3 * s[:, 0]
    assert well.sum() > 2000
    np.testing.assert_allclose(out[well], ref[well], atol=1e-10)
    # everywhere: same distance to the input (the projection may be non-unique, the distance is not)
    d_out = np.linalg.norm((out - A).reshape(n0, -1), axis=1)
    d_ref = np.linalg.norm((ref - A).reshape(n0, -1), axis=1)
    np.testing.assert_allclose(d_out, d_ref, rtol=1e-9, atol=1e-12)


def test_device_spd_solves_and_G(fixtures_dir):
    from oracle.problem import DPGOProblem
    path = os.path.join(fixtures_dir, "smallGrid3D.g2o")
    num_poses, mm = og.read_g2o_file(path)
    _, meas, _ = og.partition_measurements(num_poses, mm, 2)
    G = dpgo_amd.read_g2o(path, 2)
    opt = dpgo_amd.Options.driver()
    grp = dpgo_amd.NodeGroup(G, [0, 1], opt)
    rng = np.random.default_rng(2)
    for a in range(2):
        p = DPGOProblem(a, meas[a], opt.regularizer, LOSS_NONE, opt.reg_Cholesky_precon_max_condition_number, 0.25)
        n0 = p.n[0]
        X = rng.standard_normal((4 * n0, 3))
        out = grp.debug_apply(a, "G", X, 4 * n0)
        np.testing.assert_allclose(out, p.mat.G @ X, rtol=1e-12, atol=1e-9)
        out = grp.debug_apply(a, "solve_tt", X, 4 * n0)
        np.testing.assert_allclose(out[:n0], p.L.solve(X[:n0]), rtol=1e-9, atol=1e-9)
        out = grp.debug_apply(a, "solve_rr", X, 4 * n0)
        ref = p.precon.solve(X[n0:])
        np.testing.assert_allclose(out[n0:], ref, rtol=1e-6, atol=1e-8 * abs(ref).max())


CASES = [
    ("smallGrid3D", 2, LOSS_NONE, False, 30),    # BASELINE config 1 (MM-PGO, trivial)
    ("smallGrid3D", 2, LOSS_NONE, True, 40),
    ("smallGrid3D", 3, LOSS_HUBER, True, 40),
    ("smallGrid3D", 2, LOSS_WELSCH, True, 25),
    ("tinyGrid3D", 2, LOSS_GM, True, 15),
]


@pytest.mark.parametrize("name,nn,loss,acc,iters", CASES)
def test_trace_matches_oracle(fixtures_dir, name, nn, loss, acc, iters):
    """Per-iteration parity of the whole state machine (TNT refinement included): per-node fobj, Gk,
    gradFnorm within 1e-7 relative, poses within 1e-6 absolute, after every outer iteration."""
    orc, gpu = _pair(fixtures_dir, name, nn, loss, acc)
    for it in range(iters):
        orc.step(evaluate=False)
        assert gpu.step() == 0
        for a in range(nn):
            ro, rg = orc.nodes[a].results, gpu.group.results(a)
            assert rg.iters == ro.iters
            assert bool(rg.refined) == bool(ro.refined), (it, a)
            np.testing.assert_allclose(rg.fobj, ro.fobj[0], rtol=1e-7, err_msg="fobj it=%d node=%d" % (it, a))
            np.testing.assert_allclose(rg.Gk, ro.Gk, rtol=1e-7, err_msg="Gk it=%d node=%d" % (it, a))
            np.testing.assert_allclose(rg.gradFnorm, ro.gradFnorm, rtol=1e-5, atol=1e-7)
            assert list(rg.soft_restart_hits) == list(ro.soft_restart_hits)
            np.testing.assert_allclose(gpu.group[a].Xk(), ro.Xk, atol=1e-6, err_msg="Xk it=%d node=%d" % (it, a))
    Fo = orc.star.evaluate_f(orc.gather())
    Fg = orc.star.evaluate_f(gpu.X())
    assert abs(Fg - Fo) <= 1e-6 * abs(Fo)          # north_star: same objective within 1e-6 relative


BASELINE_CASES = [
    ("sphere2500", 1, LOSS_NONE, True, 25),     # BASELINE config 2: AMM-PGO#, num_nodes = 1
    ("torus3D", 8, LOSS_NONE, True, 20),        # BASELINE config 3
    ("city10000", 8, LOSS_NONE, True, 12),      # BASELINE config 3 (SE(2))
    ("torus3D", 8, LOSS_HUBER, True, 15),
    ("city10000", 8, LOSS_HUBER, True, 10),
]


_DIVERGED = {}   # case -> iteration at which the "refine decision flipped near its threshold" branch was taken (None: never)


@pytest.mark.parametrize("name,nn,loss,acc,iters", BASELINE_CASES)
def test_baseline_configs_match_oracle(fixtures_dir, name, nn, loss, acc, iters):
    """The BASELINE.json datasets: per-node objective trace within 1e-7 relative, final global objective
    within 1e-6 relative (north_star), rotations within 1e-6.  With one node G_tt = Laplacian + 1e-11 I is
    numerically singular along the constant vector (the gauge), so translations are compared after
    removing their mean."""
    orc, gpu = _pair(fixtures_dir, name, nn, loss, acc)
    d = orc.d
    diverged = False
    for it in range(iters):
        # the refine decision of the NEXT iterate() is a threshold test (DPGOHash.cpp:351-355); when the
        # oracle sits within 5 % of the threshold a rounding-level difference may flip it, after which the
        # two runs follow different (equally valid) trajectories to the same optimum
        near = False
        for a in range(nn):
            ro, rg = orc.nodes[a].results, gpu.group.results(a)
            ratio = ro.gradFnorm ** 2 / ro.fobj[0]
            near = near or abs(ratio / orc.options.accepted_delta - 1) < 0.05
            # ... and once converged, fobj[k] <= fobj[k-1] (restart / oscillation counters,
            # DPGOHash.cpp:181-204) is decided by rounding noise
            near = near or abs(ro.fobj[0] - ro.fobj[1]) <= 1e-9 * abs(ro.fobj[0])
        orc.step(evaluate=False)
        assert gpu.step() == 0
        for a in range(nn):
            ro, rg = orc.nodes[a].results, gpu.group.results(a)
            if bool(rg.refined) != bool(ro.refined):
                assert near, "refine decision differs away from the threshold (it=%d node=%d)" % (it, a)
                diverged = True
        if diverged:
            break
        for a in range(nn):
            ro, rg = orc.nodes[a].results, gpu.group.results(a)
            np.testing.assert_allclose(rg.fobj, ro.fobj[0], rtol=1e-7, err_msg="fobj it=%d node=%d" % (it, a))
            np.testing.assert_allclose(rg.Gk, ro.Gk, rtol=1e-7, err_msg="Gk it=%d node=%d" % (it, a))
    _DIVERGED[(name, nn, loss, acc)] = it if diverged else None
    if diverged:
        assert it >= 8      # a long common prefix was compared before the flip
        for _ in range(120):
            orc.step(evaluate=False)
            assert gpu.step() == 0
        Fo = orc.star.evaluate_f(orc.gather())
        assert abs(orc.star.evaluate_f(gpu.X()) - Fo) <= 1e-6 * abs(Fo)
        return
    for a in range(nn):
        n0 = orc.nodes[a].problem.n[0]
        Xg, Xo = gpu.group[a].Xk(), orc.nodes[a].results.Xk
        np.testing.assert_allclose(Xg[n0:n0 + d * n0], Xo[n0:n0 + d * n0], atol=1e-6)
        tg, to = Xg[:n0], Xo[:n0]
        if nn == 1:
            tg, to = tg - tg.mean(0), to - to.mean(0)
        np.testing.assert_allclose(tg, to, atol=1e-5)
    Fo = orc.star.evaluate_f(orc.gather())
    assert abs(orc.star.evaluate_f(gpu.X()) - Fo) <= 1e-6 * abs(Fo)


def test_baseline_cases_rarely_leave_the_per_iteration_comparison():
    """How often test_baseline_configs_match_oracle falls back from the per-iteration trace comparison to the final
    objective (a refine decision that flipped within 5 % of its threshold): measured on the MI355X box in round 3 --
    one of the five cases does (sphere2500 with one node, at iteration 23 of its 25: G_tt is singular along the gauge
    there and the trajectories part at rounding level; its common prefix of 23 iterations is compared at 1e-7 like the
    others), the other four are compared iteration by iteration to the end.  More than one would mean the fallback is
    hiding something."""
    if not _DIVERGED:
        pytest.skip("the baseline cases did not run in this process")
    took = {k: v for k, v in _DIVERGED.items() if v is not None}
    print("baseline cases that left the per-iteration comparison: %d of %d %s" % (len(took), len(_DIVERGED), took))
    assert len(took) <= 1, took


def test_synthetic_lattice_matches_oracle():
    """Scaled-down instance of the headline workload (same generator, Huber, AMM-PGO#, 8 nodes)."""
    from dpgo_amd import synthetic
    from oracle.g2o import Measurements
    g = synthetic.grid(12, 12, 8, 4000)
    z = np.zeros(len(g["I"]), np.int64)
    mm = Measurements(z, g["I"], z, g["J"], g["R"], g["t"], g["kappa"], g["tau"])
    G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
    X0 = G.chordal_initialization()
    orc = ODistPGO(None, 8, _oracle_opts(LOSS_HUBER, True), X0=X0, mm=mm, num_poses=g["num_poses"])
    gpu = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(LOSS_HUBER, True), X0=X0)
    for it in range(25):
        orc.step(evaluate=False)
        assert gpu.step() == 0
        for a in range(8):
            np.testing.assert_allclose(gpu.group.results(a).fobj, orc.nodes[a].results.fobj[0], rtol=1e-7,
                                       err_msg="it=%d node=%d" % (it, a))
    Fo = orc.star.evaluate_f(orc.gather())
    assert abs(orc.star.evaluate_f(gpu.X()) - Fo) <= 1e-6 * abs(Fo)
    assert abs(gpu.sum_fobj() - Fo) <= 1e-6 * abs(Fo)


def test_no_refine_path(fixtures_dir):
    """max_iterations = 0 disables the TNT branch (DPGOHash.cpp:351-355): pure proximal + solve path."""
    orc, gpu = _pair(fixtures_dir, "smallGrid3D", 2, LOSS_HUBER, True, max_iterations=0)
    for it in range(60):
        orc.step(evaluate=False)
        assert gpu.step() == 0
    for a in range(2):
        np.testing.assert_allclose(gpu.group.results(a).fobj, orc.nodes[a].results.fobj[0], rtol=1e-8)
        np.testing.assert_allclose(gpu.group[a].Xk(), orc.nodes[a].results.Xk, atol=1e-7)


def test_se2_path(fixtures_dir):
    orc, gpu = _pair(fixtures_dir, "M3500", 4, LOSS_NONE, True)
    for it in range(15):
        orc.step(evaluate=False)
        assert gpu.step() == 0
    for a in range(4):
        np.testing.assert_allclose(gpu.group.results(a).fobj, orc.nodes[a].results.fobj[0], rtol=1e-7)
    Fo = orc.star.evaluate_f(orc.gather())
    assert abs(orc.star.evaluate_f(gpu.X()) - Fo) <= 1e-6 * abs(Fo)


def test_invariants_at_scale(fixtures_dir):
    """Size-independent properties on sphere2500 / 4 nodes (no oracle trace): sum_a fobj^a == F(X)
    (Appendix B-1), rotations stay in SO(3), MM-PGO decreases F monotonically (B-3)."""
    path = os.path.join(fixtures_dir, "sphere2500.g2o")
    G = dpgo_amd.read_g2o(path, 4)
    gpu = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(LOSS_NONE, False))
    num_poses, mm = og.read_g2o_file(path)
    from oracle.star import GlobalProblem
    star = GlobalProblem(num_poses, mm, 4, OOptions.driver(LOSS_NONE, False))
    prev = np.inf
    for it in range(12):
        assert gpu.step() == 0
        X = gpu.X()
        F = star.evaluate_f(X)
        assert abs(gpu.sum_fobj() - F) <= 1e-8 * F
        assert F <= prev * (1 + 1e-12)
        prev = F
        R = X[num_poses:].reshape(num_poses, 3, 3)
        np.testing.assert_allclose(np.einsum("nij,nkj->nik", R, R), np.broadcast_to(np.eye(3), R.shape), atol=1e-12)


def test_error_conventions(fixtures_dir):
    G = dpgo_amd.read_g2o(os.path.join(fixtures_dir, "tinyGrid3D.g2o"), 2)
    grp = dpgo_amd.NodeGroup(G, [0, 1], dpgo_amd.Options.driver())
    assert grp[0].initialize(np.zeros((3, 3))) == -1            # inconsistent size -> -1
    with pytest.raises(IOError):
        dpgo_amd.read_g2o("/nonexistent.g2o", 2)


def test_multiprocess_path_matches_single_process(tmp_path):
    """bench.py with 2 ranks (gloo-staged exchange, both ranks on the one GPU of the test box) must follow
    the SAME trajectory as the single-process run: the pack / all-gather / unpack path is exact."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--grid", "10,10,8,2400", "--steps", "6", "--warmup", "2", "--no-prof", "--no-cpu", "--converge", "30"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    # (a) under the launcher the driver uses, (b) started directly: bench.py spawns its own two ranks
    for cmd in ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                 "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py")],
                [sys.executable, os.path.join(root, "bench.py")]):
        two = subprocess.run(cmd + ["--gpus", "2", "--backend", "gloo", "--share-gpu"] + common,
                             capture_output=True, text=True, timeout=900, cwd=root, env=env)
        assert two.returncode == 0, two.stderr[-3000:]
        lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        j2 = json.loads(lines[0])
        assert j2["n_gpus"] == 2 and j2["config"]["nodes_per_gpu"] == 4
        assert abs(j1["objective_2F"] - j2["objective_2F"]) <= 1e-10 * abs(j1["objective_2F"])
        # round 5: the N > 1 line carries the second half of the metric too -- the same trajectory, so the same
        # objectives at the same iterations as the one-rank line (the clocks differ)
        c1, c2 = j1["convergence"], j2["convergence"]
        assert c2["iterations_run"] == 30 and c2["iterations_to_1e-6"] == c1["iterations_to_1e-6"]
        assert abs(c1["lowest_2F"] - c2["lowest_2F"]) <= 1e-10 * abs(c1["lowest_2F"])
        assert c2["seconds_to_1e-6"] > 0 and abs(j2["iters_per_s_to_objective"] - c2["iterations_to_1e-6"] / c2["seconds_to_1e-6"]) < 1e-6 * j2["iters_per_s_to_objective"]
    # round 6: the unpack is lazy -- update()'s inter-edge pass reads the neighbour rows out of the gathered buffer and stores
    # them into Xk on the way (Group::set_pending_recv); with the plain unpack kernel (DPGO_LAZY_UNPACK=0) the same bits
    two0 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu"] + common,
                          capture_output=True, text=True, timeout=900, cwd=root, env=dict(env, DPGO_LAZY_UNPACK="0"))
    assert two0.returncode == 0, two0.stderr[-3000:]
    j0 = json.loads([l for l in two0.stdout.splitlines() if l.startswith("{")][-1])
    assert j0["objective_2F"] == j2["objective_2F"] and j0["convergence"]["lowest_2F"] == j2["convergence"]["lowest_2F"]
    # ... and with both ranks' hosts 150 us late to every read-back (the streams run that far ahead of them): the same bits
    late = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu"] + common,
                          capture_output=True, text=True, timeout=900, cwd=root, env=dict(env, DPGO_DEBUG_LATE_HOST_US="150"))
    assert late.returncode == 0, late.stderr[-3000:]
    jl = json.loads([l for l in late.stdout.splitlines() if l.startswith("{")][-1])
    assert jl["objective_2F"] == j2["objective_2F"] and jl["convergence"]["lowest_2F"] == j2["convergence"]["lowest_2F"]


@pytest.mark.parametrize("loss", [LOSS_NONE, LOSS_HUBER])
def test_global_objective_and_gradient_from_device_state(fixtures_dir, loss):
    """Row a14: the driver's log line (2F, 2|grad F|) from the per-node device reductions equals
    DPGOStar::evaluate_f / evaluate_grad of the oracle on the gathered X (1e-6 relative; the robust
    gradient differs from the oracle's B-form at the level of the reference's own kappa*I vs
    kappa*R R^T inconsistency, 1e-6)."""
    orc, gpu = _pair(fixtures_dir, "torus3D", 8, loss, True)
    for it in range(6):
        assert gpu.step() == 0
        X = gpu.X()
        F2, g2 = gpu.evaluate()
        Fo = 2 * orc.star.evaluate_f(X)
        go = 2 * float(np.linalg.norm(orc.star.evaluate_grad(X)))
        assert abs(F2 - Fo) <= 1e-6 * Fo
        assert abs(g2 - go) <= 1e-5 * go


def test_per_node_calls_equal_batched_calls(fixtures_dir):
    """DPGOHash-style per-node update()/iterate() (device masks selecting one node at a time, as the
    reference driver's for-alpha loops do) gives the same state as the batched group calls."""
    path = os.path.join(fixtures_dir, "smallGrid3D.g2o")
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    G = dpgo_amd.read_g2o(path, 3)
    opt = dpgo_amd.Options.driver(LOSS_HUBER, True)
    a = dpgo_amd.DistPGO(G, opt, X0=X0)
    b = dpgo_amd.DistPGO(G, opt, X0=X0)
    for it in range(12):
        assert a.step() == 0
        for k in range(3):
            assert b.group[k].iterate() == 0
        assert b.group.communicate_local() == 0
        for k in reversed(range(3)):
            assert b.group[k].update() == 0
        for k in range(3):
            ra, rb = a.group.results(k), b.group.results(k)
            assert ra.iters == rb.iters and ra.fobj == rb.fobj and ra.Gk == rb.Gk
            np.testing.assert_array_equal(a.group[k].Xk(), b.group[k].Xk())
    # iterate() before update() is an error, as in the reference (assert in DPGOHash.cpp:233)
    assert b.group[0].iterate() == 0
    assert b.group[0].iterate() == -1


def test_replayed_cg_steps_serve_a_strict_subset_of_the_group(fixtures_dir, tmp_path):
    """Per-node iterate() calls refine ONE node of an eight-node group at a time (city10000: several CG steps per refinement).  With replays forced, the CG steps of that
    node go out as replays of a graph whose by-value node set is the whole group -- the device's own masks keep the other
    seven nodes out -- while everything else of a partial node set is launched eagerly (Group::segment).  Bit for bit the
    trajectory of eager launches, and of the batched calls."""
    import subprocess
    import sys
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import dpgo_amd
G = dpgo_amd.read_g2o(%r, 8)
drv = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(0, True))
per_node = sys.argv[2] == "1"
steps = 0
for it in range(25):
    if per_node:
        for k in range(8):
            assert drv.group[k].iterate() == 0
        assert drv.group.communicate_local() == 0
        for k in range(8):
            assert drv.group[k].update() == 0
    else:
        assert drv.step() == 0
    steps += sum(int(drv.group.results(k).tnt_inner_iterations) for k in range(8) if drv.group.results(k).refined)
np.save(sys.argv[1], drv.X())
print("STATS", drv.group.graph_stats(), steps)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(fixtures_dir, "city10000.g2o"))

    def run(tag, per_node, **env):
        path = str(tmp_path / (tag + ".npy"))
        out = subprocess.run([sys.executable, "-c", code, path, "1" if per_node else "0"], env=dict(os.environ, **env),
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        stats = [l for l in out.stdout.splitlines() if l.startswith("STATS")][-1]
        return np.load(path), stats

    base, _ = run("batched_eager", False, DPGO_ITER_GRAPH="0")
    got, stats = run("per_node_replayed", True, DPGO_ITER_GRAPH="1")
    assert np.array_equal(got, base)
    assert "'replays': 0" not in stats, stats          # (the per-node run did replay: its CG steps)
    got, _ = run("per_node_eager", True, DPGO_ITER_GRAPH="0")
    assert np.array_equal(got, base)


def test_long_runs_replayed_or_switching_midway_equal_eager_runs(fixtures_dir, tmp_path):
    """300 iterations of city10000 / 8 nodes / Huber -- through the early regime, the interior one (several CG steps per
    refinement, nodes refining and not, rejected steps, restarts) and past convergence -- (i) eagerly, (ii) with every
    segment replayed from the first iteration (the graph cache fills, evicts and hits its capture cap), (iii) starting
    eagerly and switching to replays after 32 iterations, as a group does once its host is found to be the slower side
    (DPGO_HOST_BOUND_BELOW=2: every host counts as slower).  Bit for bit the same trajectory."""
    import subprocess
    import sys
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import dpgo_amd
G = dpgo_amd.read_g2o(%r, 8)
drv = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(1, True))
trace = []
for it in range(300):
    assert drv.step() == 0
    trace.append(drv.sum_fobj())
np.savez(sys.argv[1], X=drv.X(), trace=np.asarray(trace))
print("STATS", drv.group.graph_stats())
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(fixtures_dir, "city10000.g2o"))

    def run(tag, **env):
        path = str(tmp_path / (tag + ".npz"))
        base_env = {k: v for k, v in os.environ.items() if k not in ("DPGO_ITER_GRAPH", "DPGO_HOST_BOUND_BELOW")}
        out = subprocess.run([sys.executable, "-c", code, path], env=dict(base_env, **env), capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        stats = eval([l for l in out.stdout.splitlines() if l.startswith("STATS")][-1][6:])
        return np.load(path), stats

    eager, st = run("eager", DPGO_ITER_GRAPH="0")
    assert st["replays"] == 0
    for tag, env in (("replayed", dict(DPGO_ITER_GRAPH="1")), ("switching", dict(DPGO_HOST_BOUND_BELOW="2"))):
        got, st = run(tag, **env)
        assert st["replays"] > 300 and st["captures"] >= 4, (tag, st)
        assert np.array_equal(got["trace"], eager["trace"]), tag
        assert np.array_equal(got["X"], eager["X"]), tag
        if tag == "switching":
            assert st["eager"] >= 32, st            # (the first 32 iterations ran their segments eagerly)


def test_dist_pgo_cli_matches_oracle(fixtures_dir, tmp_path):
    """The C++ driver (reference flags / stdout / result files, dist_pgo.cpp:23-47, 493-568)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "dpgo_amd", "dist_pgo")
    path = os.path.join(fixtures_dir, "smallGrid3D.g2o")
    out = subprocess.run([exe, "--dataset", path, "--num_nodes", "2", "--iters", "25", "--loss", "huber",
                          "--dist_init", "false"], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l[:1].isdigit() and ": " in l]
    assert len(lines) == 25 and lines[0].startswith("0: ")
    final = float([l for l in out.stdout.splitlines() if l.startswith("final objective")][0].split(":")[1])
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    orc = ODistPGO(path, 2, _oracle_opts(LOSS_HUBER, True), X0=X0, mm=mm, num_poses=num_poses)
    # the driver's chordal init is its own (host PCG): compare the iteration-0 line with the oracle at ITS X0 loosely,
    # and the converged objective tightly
    f0 = float(lines[0].split()[1])
    assert abs(f0 - orc.trace[0][0]) <= 1e-6 * orc.trace[0][0]
    orc.run(25)
    assert abs(final - orc.trace[-1][0]) <= 1e-6 * orc.trace[-1][0]
    res = np.loadtxt(os.path.join(tmp_path, "results_chordal_2_amm.txt"))
    assert res.shape == (26, 4) and res[-1, 0] == 25 and abs(res[-1, 2] - final) <= 1e-9 * final
    est = np.loadtxt(os.path.join(tmp_path, "estimates_huber.txt"))
    assert est.shape == (4 * num_poses, 3)
    np.testing.assert_allclose(est[0], 0, atol=1e-12)                      # t_0 = 0 after the gauge fix
    np.testing.assert_allclose(est[num_poses:num_poses + 3], np.eye(3), atol=1e-5)   # R_0 = I


@pytest.mark.parametrize("name,nn,loss,iters", [("M3500", 4, LOSS_NONE, 25),      # BASELINE config 5 (AMM-PGO*)
                                                 ("smallGrid3D", 3, LOSS_HUBER, 25),
                                                 ("torus3D", 8, LOSS_WELSCH, 12)])
def test_amm_pgo_star_matches_oracle(fixtures_dir, name, nn, loss, iters):
    """AMM-PGO* (DPGOStar::iterate): global objective after every iteration within 1e-7 relative, same
    branch decisions, final poses within 1e-6."""
    from oracle.star import DPGOStar as OStar
    path = os.path.join(fixtures_dir, name + ".g2o")
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    orc = OStar(path, nn, _oracle_opts(loss, True), mm=mm, num_poses=num_poses)
    orc.initialize(X0)
    gpu = dpgo_amd.DPGOStar(dpgo_amd.read_g2o(path, nn), dpgo_amd.Options.driver(loss, True))
    assert gpu.initialize(X0) == 0
    np.testing.assert_allclose(gpu.state()["fobj"], orc.fobj, rtol=1e-9)
    names = {"pm": 1, "mm": 2, "phi": 4}
    for it in range(iters):
        orc.step()
        assert gpu.step() == 0
        st = gpu.state()
        np.testing.assert_allclose(st["fobj"], orc.fobj, rtol=1e-7, err_msg="it=%d" % it)
        np.testing.assert_allclose(st["fobjh"], orc.fobjh, rtol=1e-7, err_msg="it=%d" % it)
        np.testing.assert_allclose(st["F"], orc.F, rtol=1e-9)
        assert st["branches"] == sum(names[b] for b in orc.branches), (it, orc.branches)
    # poses: SURVEY 8d parity bar (translations 1e-5 x scale, rotations 1e-5 rad); the objectives above are the
    # sharp comparison -- with G_tt regularised by 1e-11 the translations move at the 1e-6 level with the
    # elimination order of the solver
    np.testing.assert_allclose(gpu.X(), orc.Xk, atol=1e-5)
    # iterate() without update() is an error
    assert gpu.iterate() == -1


def test_receive_messages_equal_communicate(fixtures_dir):
    """DPGOHash::receive with per-neighbour messages (DPGOHash.cpp:45-82) fills the same neighbour rows as
    communicate() (DPGOHash.h:28-86)."""
    path = os.path.join(fixtures_dir, "torus3D.g2o")
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    G = dpgo_amd.read_g2o(path, 4)
    opt = dpgo_amd.Options.driver(LOSS_NONE, True)
    a = dpgo_amd.DistPGO(G, opt, X0=X0)
    b = dpgo_amd.DistPGO(G, opt, X0=X0)
    for it in range(3):
        assert a.step() == 0
        assert b.group.iterate() == 0
        msgs = {k: {} for k in range(4)}
        for k in range(4):
            for beta in range(4):
                M = b.group[k].message_for(beta)
                if M is not None:
                    msgs[beta][k] = M
        for k in range(4):
            assert b.group[k].receive(msgs[k]) == 0
        assert b.group.update() == 0
        for k in range(4):
            np.testing.assert_array_equal(a.group[k].Xk(), b.group[k].Xk())
            assert a.group.results(k).fobj == b.group.results(k).fobj
    assert b.group[0].receive({0: np.zeros((4, 3))}) == -1      # not a neighbour -> error, as LOG(ERROR) at :77


def test_cpp_facade_runs_the_driver_loop(fixtures_dir, golden_dir):
    """examples/facade_mm.cpp = the reference driver loop on include/dpgo_amd.hpp (DPGO::DPGOHashGroup): its stdout
    trace equals the golden oracle trace of BASELINE config 1 (smallGrid3D, MM-PGO, 2 nodes, 200 iterations); the
    only difference is the chordal initialisation (host PCG vs. sparse direct), hence 1e-6."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "dpgo_amd", "facade_mm")
    assert os.path.exists(exe), "build with __graft_entry__.build()"
    out = subprocess.run([exe, os.path.join(fixtures_dir, "smallGrid3D.g2o"), "2", "200", "trivial", "0"], check=True,
                         capture_output=True, text=True).stdout.strip().split("\n")
    got = np.array([[float(v) for v in line.split(":")[1].split()] for line in out])
    with open(os.path.join(golden_dir, "oracle_traces.json")) as fh:
        ref = np.asarray(json.load(fh)["cases"]["config1_smallGrid3D_mm_2nodes"]["trace_2F_2gradnorm"])
    assert got.shape == ref.shape
    np.testing.assert_allclose(got[:, 0], ref[:, 0], rtol=1e-6)
    np.testing.assert_allclose(got[:, 1], ref[:, 1], rtol=1e-3, atol=1e-6 * ref[0, 1])


def test_runs_are_bit_reproducible(fixtures_dir):
    """No atomics anywhere on the data path: partial sums are combined in a fixed order, the solve pulls its
    children's contributions in list order.  Two runs of the same problem give bit-identical iterates."""
    path = os.path.join(fixtures_dir, "torus3D.g2o")
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    outs = []
    for _ in range(2):
        G = dpgo_amd.read_g2o(path, 8)
        drv = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(LOSS_HUBER, True), X0=X0)
        for _ in range(15):
            assert drv.step() == 0
        outs.append(drv.X().copy())
    assert np.array_equal(outs[0], outs[1])


def test_native_step_equals_the_three_calls(fixtures_dir):
    """dpgo_group_step (the loop body of dist_pgo.cpp:496-521 in one call) is iterate + communicate + update: the same
    launches in the same order, so the same bits."""
    path = os.path.join(fixtures_dir, "torus3D.g2o")
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    outs = []
    for native in (True, False):
        G = dpgo_amd.read_g2o(path, 8)
        drv = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(LOSS_HUBER, True), X0=X0)
        for _ in range(12):
            if native:
                assert drv.step() == 0
            else:
                g = drv.group
                assert g.iterate() == 0 and g.communicate_local() == 0 and g.update() == 0
        outs.append(drv.X().copy())
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("kw", [
    dict(max_iterations_accepted=3, max_iterations=12),          # several accepted TNT steps: the model is rebuilt
    dict(preconditioner=0),                                      # Preconditioner::None (identity)
    dict(preconditioner=1),                                      # Preconditioner::Jacobi (diag(G_RR)^-1)
    dict(preconditioner=1, max_iterations_accepted=2, max_iterations=8),
    dict(max_iterations_accepted=2, max_tCG_iterations=3),       # truncated CG that stops on its iteration cap
])
@pytest.mark.parametrize("loss", [LOSS_NONE, LOSS_HUBER])
def test_tnt_option_variants_match_oracle(fixtures_dir, loss, kw):
    """TNT / STPCG options other than the driver's (DPGO_types.h:78-201): the refinement paths the default run never
    takes (a second accepted step re-using the trial point's model gradient, no preconditioner, capped CG)."""
    orc, gpu = _pair(fixtures_dir, "smallGrid3D", 2, loss, True, **kw)
    for it in range(12):
        orc.step(evaluate=False)
        assert gpu.step() == 0
        for a in range(2):
            ro, rg = orc.nodes[a].results, gpu.group.results(a)
            np.testing.assert_allclose(rg.fobj, ro.fobj[0], rtol=1e-7, err_msg="fobj it=%d node=%d %r" % (it, a, kw))
            np.testing.assert_allclose(rg.Gk, ro.Gk, rtol=1e-7, err_msg="Gk it=%d node=%d %r" % (it, a, kw))
    np.testing.assert_allclose(gpu.X(), orc.gather(), atol=1e-6)


def _random_global_X(rng, N, d, scale=3.0):
    X = np.zeros(((d + 1) * N, d))
    X[:N] = scale * rng.standard_normal((N, d))
    q, _ = np.linalg.qr(rng.standard_normal((N, d, d)))
    q[:, :, 0] *= np.sign(np.linalg.det(q))[:, None]
    X[N:] = q.transpose(0, 2, 1).reshape(N * d, d)
    return X


@pytest.mark.parametrize("name,nn", [("smallGrid3D", 2), ("M3500", 4), ("torus3D", 3)])
@pytest.mark.parametrize("loss", [LOSS_NONE, LOSS_HUBER, LOSS_GM, LOSS_WELSCH])
def test_evaluate_f_and_grad_at_arbitrary_X(fixtures_dir, name, nn, loss):
    """dpgo_group_evaluate == DPGOStar::evaluate_f / evaluate_grad (DPGOStar.cpp:713-829) at RANDOM points (not
    linearisation points), through the C ABI; the optimizer state is not touched.  1e-11 relative."""
    path = os.path.join(fixtures_dir, name + ".g2o")
    num_poses, mm = og.read_g2o_file(path)
    from oracle.star import GlobalProblem
    star = GlobalProblem(num_poses, mm, nn, _oracle_opts(loss, True))
    G = dpgo_amd.read_g2o(path, nn)
    X0 = chordal_initialization(num_poses, mm)
    gpu = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(loss, True), X0=X0)
    ref = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(loss, True), X0=X0)
    rng = np.random.default_rng(7)
    for X in (_random_global_X(rng, num_poses, G.d), X0, _random_global_X(rng, num_poses, G.d, 0.1)):
        F, g2, grad = gpu.group.evaluate(X, want_grad=True)
        Fo, go = star.evaluate_f(X), star.evaluate_grad(X)
        # trivial loss: the reference's 1/2 tr(X^T M X) cancels |x|^2-sized terms down to residual size (M3500: 1e-10
        # of F is lost there); the device sums squared residuals, so the comparison is only as good as the oracle
        assert abs(F - Fo) <= (1e-9 if loss == LOSS_NONE else 1e-11) * abs(Fo)
        np.testing.assert_allclose(grad, go, rtol=0, atol=1e-10 * np.abs(go).max())
        assert abs(g2 - np.sum(go * go)) <= 1e-9 * np.sum(go * go)
        assert gpu.step() == 0 and ref.step() == 0                     # evaluating leaves the trajectory alone
        assert np.array_equal(gpu.X(), ref.X())


def test_set_options_and_accessors(fixtures_dir):
    """DPGOHash::set_options (DPGOHash.h:93-96): optimizer-level fields change the run exactly as a group created
    with them; problem-level fields are refused."""
    path = os.path.join(fixtures_dir, "smallGrid3D.g2o")
    G = dpgo_amd.read_g2o(path, 2)
    X0 = G.chordal_initialization()
    a = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(LOSS_HUBER, True), X0=X0)
    b = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(LOSS_HUBER, True, max_iterations_accepted=3, max_iterations=12), X0=X0)
    o = dpgo_amd.Options.driver(LOSS_HUBER, True, max_iterations_accepted=3, max_iterations=12)
    assert a.group.set_options(o) == 0
    got = a.group.get_options()
    assert (got.max_iterations_accepted, got.max_iterations, got.rescale) == (3, 12, dpgo_amd.RESCALE_STATIC)
    for _ in range(6):
        assert a.step() == 0 and b.step() == 0
    assert np.array_equal(a.X(), b.X())
    assert a.group.set_options(dpgo_amd.Options.driver(LOSS_NONE, True)) == -1      # the loss is part of the problem
    assert a.group.set_options(dpgo_amd.Options.driver(LOSS_HUBER, True, regularizer=1e-3)) == -1
    assert a.group.update([0, 0]) == -1 and a.group.iterate([5]) == -1             # locals are validated
    with pytest.raises(RuntimeError):
        dpgo_amd.NodeGroup(G, [0, 1], dpgo_amd.Options.driver(LOSS_HUBER, True, preconditioner=dpgo_amd.PRECON_ICHOL))


@pytest.mark.parametrize("loss", [LOSS_NONE, LOSS_HUBER])
def test_update_twice_in_one_iteration_keeps_history(fixtures_dir, loss):
    """update -> receive() (clears `updated`) -> update at the same iteration: X[iter-1], g[iter-1], fobj[iter-1],
    s[iter] stay those of the first call (the reference overwrites X[iter] in place, DPGOHash.cpp:99-106), the
    restart counters run again.  Oracle = the same call sequence on the restatement."""
    path = os.path.join(fixtures_dir, "smallGrid3D.g2o")
    orc, gpu = _pair(fixtures_dir, "smallGrid3D", 2, loss, True)
    for it in range(8):
        for nd in orc.nodes:
            nd.iterate()
        for nd in orc.nodes:
            nd.communicate(orc.nodes)
        for nd in orc.nodes:
            nd.update()
        assert gpu.group.iterate() == 0 and gpu.group.communicate_local() == 0 and gpu.group.update() == 0
        if it in (0, 3, 4):
            # every node re-receives what its neighbour owes it (same numbers), then updates again
            for k in range(2):
                other = 1 - k
                n_o = orc.nodes[other]
                sent = n_o.problem.info.sent[k]
                rows = [v[1] for v in sent.values()]
                n0o, d = n_o.problem.n[0], n_o.problem.d
                M = np.vstack([n_o.results.Xk[rows]] + [n_o.results.Xk[n0o + r * d: n0o + r * d + d] for r in rows])
                assert orc.nodes[k].receive({other: M}) == 0
                assert gpu.group[k].receive({other: gpu.group[other].message_for(k)}) == 0
            for nd in orc.nodes:
                nd.update()
            assert gpu.group.update() == 0
        for a in range(2):
            ro, rg = orc.nodes[a].results, gpu.group.results(a)
            np.testing.assert_allclose(rg.fobj, ro.fobj[0], rtol=1e-7, err_msg="it=%d" % it)
            np.testing.assert_allclose(rg.gamma, ro.gamma, rtol=1e-12)
            np.testing.assert_allclose([rg.Fk[0], rg.Fk[1]], ro.Fk, rtol=1e-7)
            assert list(rg.soft_restart_hits) == list(ro.soft_restart_hits)
    np.testing.assert_allclose(gpu.X(), orc.gather(), atol=1e-6)


def test_headline_size_properties():
    """The full 100 000-pose / 400 000-edge headline graph (BASELINE config 4), a few iterations: properties that do
    not need the oracle: sum_a fobj^a = F(X) from the independent k_cost pass at 1e-8, rotations
    in SO(3), the translation system G_tt t + G_tR R + g_t = 0 solved to 1e-9 of its scale, objective decreasing over
    the window, and two runs bit-identical."""
    from dpgo_amd import synthetic
    g = synthetic.grid(50, 50, 40, 400000, seed=synthetic.HEADLINE["seed"])
    N = g["num_poses"]
    G = dpgo_amd.graph_from_edges(3, N, g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
    X0 = G.chordal_initialization()
    opt = dpgo_amd.Options.driver(LOSS_HUBER, True)
    grp = dpgo_amd.NodeGroup(G, range(8), opt)
    outs = []
    for run in range(2):
        assert grp.initialize_global(X0) == 0 and grp.update() == 0
        F0 = sum(grp.results(k).fobj for k in range(8))
        for it in range(4):
            assert grp.iterate() == 0 and grp.communicate_local() == 0 and grp.update() == 0
        X = np.zeros((4 * N, 3), order="F")
        grp.scatter_global(X)
        outs.append(X.copy())
        if run:
            break
        Fsum = sum(grp.results(k).fobj for k in range(8))
        F, g2 = grp.evaluate(X)
        assert abs(Fsum - F) <= 1e-8 * F, (Fsum, F)
        assert abs(sum(grp.results(k).gradFnorm ** 2 for k in range(8)) - g2) <= 1e-8 * g2
        assert F < F0
        R = X[N:].reshape(N, 3, 3)
        np.testing.assert_allclose(np.einsum("nij,nkj->nik", R, R), np.broadcast_to(np.eye(3), R.shape), atol=1e-12)
        np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-12)
        # exactness of the multifrontal solve at this size: residual of G_tt t = b against the device operator
        n0 = G.node_sizes(3)[0]
        rng = np.random.default_rng(3)
        B = np.zeros((4 * n0, 3))
        B[:n0] = rng.standard_normal((n0, 3))
        T = grp.debug_apply(3, "solve_tt", B, 4 * n0)
        Z = np.zeros((4 * n0, 3))
        Z[:n0] = T[:n0]
        GT = grp.debug_apply(3, "G", Z, 4 * n0)
        assert np.abs(GT[:n0] - B[:n0]).max() <= 1e-9 * np.abs(B[:n0]).max()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("name,nn,loss,acc,iters", [
    ("smallGrid3D", 2, LOSS_HUBER, True, 40), ("smallGrid3D", 2, LOSS_GM, True, 25), ("smallGrid3D", 2, LOSS_WELSCH, False, 25),
    ("tinyGrid3D", 2, LOSS_HUBER, True, 30), ("M3500", 4, LOSS_HUBER, True, 25), ("torus3D", 3, LOSS_WELSCH, True, 15)])
def test_dynamic_rescale_matches_oracle(fixtures_dir, name, nn, loss, acc, iters):
    """Rescale::Dynamic, the default of DPGO::Options (DPGO_types.h:128; evaluate_g_and_f*_rescale +
    update_quadratic_mat, DPGOProblem.cpp:289-358, 426-514, 751-840): every inter-node edge carries a scale that follows
    its loss weight; per-iteration traces against the oracle."""
    orc, gpu = _pair(fixtures_dir, name, nn, loss, acc, rescale=1)
    assert gpu.group.get_options().rescale == dpgo_amd.RESCALE_DYNAMIC
    rescaled = 0
    for it in range(iters):
        orc.step(evaluate=False)
        assert gpu.step() == 0
        for a in range(nn):
            ro, rg = orc.nodes[a].results, gpu.group.results(a)
            np.testing.assert_allclose(rg.fobj, ro.fobj[0], rtol=1e-7, err_msg="fobj it=%d node=%d" % (it, a))
            np.testing.assert_allclose(rg.Gk, ro.Gk, rtol=1e-7, err_msg="Gk it=%d node=%d" % (it, a))
            rescaled += int(ro.rescale_count == 0)
    assert rescaled > 0                                   # the surrogate was rescaled along the way
    if name != "M3500":                                  # (M3500 has no outliers: every weight stays 1)
        assert min(nd.problem.scale.min() for nd in orc.nodes) < 1.0
    # poses: SURVEY 8d states 1e-5 (rad / scale).  The two sides estimate lambda_max of G_RR independently to 1e-4 (ARPACK
    # with its process-wide random start on the oracle's side, Lanczos on the library's), so the preconditioner and with
    # it the truncated CG differ in the last digits: 1.1e-6 on two rotation entries of M3500 after 25 iterations
    np.testing.assert_allclose(gpu.X(), orc.gather(), atol=5e-6)
    Fo = orc.star.evaluate_f(orc.gather())
    assert abs(gpu.sum_fobj() - Fo) <= 1e-6 * abs(Fo)


def test_dynamic_rescale_with_amm_pgo_star(fixtures_dir):
    """AMM-PGO* re-linearises with evaluate_g_and_f0_rescale every iteration (DPGOStar.cpp:350-355)."""
    from oracle.star import DPGOStar as ODPGOStar
    path = os.path.join(fixtures_dir, "smallGrid3D.g2o")
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    orc = ODPGOStar(path, 2, _oracle_opts(LOSS_HUBER, True, rescale=1), mm=mm, num_poses=num_poses)
    orc.initialize(X0)
    star = dpgo_amd.DPGOStar(dpgo_amd.read_g2o(path, 2), dpgo_amd.Options.driver(LOSS_HUBER, True, rescale=1))
    assert star.initialize(X0) == 0
    for it in range(20):
        orc.step()
        assert star.step() == 0
        np.testing.assert_allclose(star.state()["fobj"], orc.fobj, rtol=1e-7, err_msg="it=%d" % it)


def test_projection_kernel_matches_reference_avx_vectors(fixtures_dir, golden_dir):
    """The device projection (project_so3 / project_so2 in kernels.hip) against the outputs of the REFERENCE's own AVX2
    kernels (tests/golden/so_ref.npz, see tests/test_oracle_so.py): reference-pinned parity for row a9."""
    z = np.load(os.path.join(golden_dir, "so_ref.npz"))
    from tests.test_oracle_so import well_conditioned
    G = dpgo_amd.read_g2o(os.path.join(fixtures_dir, "sphere2500.g2o"), 1)
    grp = dpgo_amd.NodeGroup(G, [0], dpgo_amd.Options.driver())
    n0, A, U = 2500, z["A3"], z["U3"]
    pad = np.tile(np.eye(3), (n0 - len(A), 1, 1))
    out = grp.debug_apply(0, "project", np.concatenate([A, pad]).reshape(3 * n0, 3), 3 * n0).reshape(n0, 3, 3)[:len(A)]
    well = well_conditioned(A)
    np.testing.assert_allclose(out[well], U[well], rtol=0, atol=1e-9)
    d_out = np.linalg.norm((out - A).reshape(len(A), -1), axis=1)
    d_ref = np.linalg.norm((U - A).reshape(len(A), -1), axis=1)
    np.testing.assert_allclose(d_out, d_ref, rtol=1e-8, atol=1e-12)
    # rotations to rounding, whatever the input (the sweeps stop early per pose, the rotation's cosine comes from rsqrt)
    np.testing.assert_allclose(np.einsum("nij,nkj->nik", out, out), np.tile(np.eye(3), (len(A), 1, 1)), rtol=0, atol=4e-15)
    np.testing.assert_allclose(np.linalg.det(out), 1.0, rtol=0, atol=4e-15)
    G2 = dpgo_amd.read_g2o(os.path.join(fixtures_dir, "M3500.g2o"), 1)
    grp2 = dpgo_amd.NodeGroup(G2, [0], dpgo_amd.Options.driver())
    n2, A2, U2 = 3500, z["A2"], z["U2"]
    pad = np.tile(np.eye(2), (n2 - len(A2), 1, 1))
    out2 = grp2.debug_apply(0, "project", np.concatenate([A2, pad]).reshape(2 * n2, 2), 2 * n2).reshape(n2, 2, 2)[:len(A2)]
    np.testing.assert_allclose(out2, U2, rtol=0, atol=1e-15)


def test_dynamic_rescale_on_synthetic_lattice():
    """Rescale::Dynamic on the scaled-down headline workload (outlier closures: weights well below 1, a rescale of some
    nodes -- not all -- in most iterations: the narrower second pass over the partial sums must leave the other nodes' alone)."""
    from dpgo_amd import synthetic
    from oracle.g2o import Measurements
    g = synthetic.grid(12, 12, 8, 4000)
    z = np.zeros(len(g["I"]), np.int64)
    mm = Measurements(z, g["I"], z, g["J"], g["R"], g["t"], g["kappa"], g["tau"])
    G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
    X0 = G.chordal_initialization()
    orc = ODistPGO(None, 8, _oracle_opts(LOSS_HUBER, True, rescale=1), X0=X0, mm=mm, num_poses=g["num_poses"])
    gpu = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(LOSS_HUBER, True, rescale=1), X0=X0)
    for it in range(30):
        orc.step(evaluate=False)
        assert gpu.step() == 0
        for a in range(8):
            np.testing.assert_allclose(gpu.group.results(a).fobj, orc.nodes[a].results.fobj[0], rtol=1e-7,
                                       err_msg="it=%d node=%d" % (it, a))
    Fo = orc.star.evaluate_f(orc.gather())
    assert abs(orc.star.evaluate_f(gpu.X()) - Fo) <= 1e-6 * abs(Fo)
    # (with Dynamic rescale fobj follows the reference's recursion fobj = Gk - ... (DPGOProblem.cpp:426-514), which
    # tracks F to 1e-5 here, not to rounding: the sum is compared with the oracle's own sum)
    So = sum(nd.results.fobj[0] for nd in orc.nodes)
    assert abs(gpu.sum_fobj() - So) <= 1e-7 * abs(So)
    assert abs(So - Fo) <= 1e-4 * abs(Fo)


@pytest.mark.parametrize("loss,acc,kw", [
    (LOSS_NONE, False, {}),                         # MM-PGO, trivial loss
    (LOSS_NONE, True, {}),
    (LOSS_GM, True, {}),
    (LOSS_WELSCH, False, {}),
    (LOSS_HUBER, True, dict(preconditioner=0)),
    (LOSS_HUBER, True, dict(preconditioner=1)),
    (LOSS_HUBER, True, dict(max_iterations_accepted=3, max_iterations=12)),
])
def test_option_matrix_at_scale(loss, acc, kw):
    """Every scheme / loss / TNT variant on a 32 x 32 x 24 lattice (24 576 poses, 6 nodes of unequal boundary sizes, the
    tile classes of the big factor) without the oracle: sum_a fobj^a = F from the independent cost pass, |grad| from the
    per-node norms, rotations in SO(3), MM monotone, the objective below its start."""
    from dpgo_amd import synthetic
    g = synthetic.grid(32, 32, 24, 98304)
    N = g["num_poses"]
    G = dpgo_amd.graph_from_edges(3, N, g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 6)
    X0 = G.chordal_initialization()
    grp = dpgo_amd.NodeGroup(G, range(6), dpgo_amd.Options.driver(loss, acc, **kw))
    assert grp.initialize_global(X0) == 0 and grp.update() == 0
    F0 = prev = sum(grp.results(k).fobj for k in range(6))
    for it in range(8):
        assert grp.iterate() == 0 and grp.communicate_local() == 0 and grp.update() == 0
        Fsum = sum(grp.results(k).fobj for k in range(6))
        assert np.isfinite(Fsum)
        if not acc:
            assert Fsum <= prev * (1 + 1e-12), (it, Fsum, prev)      # MM-PGO is monotone
        prev = Fsum
    X = np.zeros((4 * N, 3), order="F")
    grp.scatter_global(X)
    F, g2 = grp.evaluate(X)
    assert abs(Fsum - F) <= 1e-8 * F, (Fsum, F)
    assert abs(sum(grp.results(k).gradFnorm ** 2 for k in range(6)) - g2) <= 1e-8 * g2
    assert F < F0
    R = X[N:].reshape(N, 3, 3)
    np.testing.assert_allclose(np.einsum("nij,nkj->nik", R, R), np.broadcast_to(np.eye(3), R.shape), atol=1e-12)


@pytest.mark.parametrize("loss", [LOSS_NONE, LOSS_HUBER])
def test_amm_pgo_star_at_scale(loss):
    """AMM-PGO* (BASELINE config 5's scheme) on the 24 576-pose lattice, 4 nodes, no oracle: the master's fobj
    is the global objective at the iterate (independent cost pass, 1e-8) and its reference value F never increases
    (DPGOStar.cpp:126-213: a step that does not decrease it is taken again without extrapolation)."""
    from dpgo_amd import synthetic
    g = synthetic.grid(32, 32, 24, 98304)
    N = g["num_poses"]
    G = dpgo_amd.graph_from_edges(3, N, g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 4)
    star = dpgo_amd.DPGOStar(G, dpgo_amd.Options.driver(loss, True))
    assert star.initialize(G.chordal_initialization()) == 0
    prev = None
    for it in range(8):
        assert star.step() == 0
        F = star.state()["F"]
        assert np.isfinite(F)
        if prev is not None:
            assert F <= prev * (1 + 1e-12), (it, F, prev)
        prev = F
    Fx, _ = star.group.evaluate(star.X())
    assert star.update() == 0                      # fobj of the state refers to the last linearisation point
    s = star.state()
    assert abs(s["fobj"] - Fx) <= 1e-8 * Fx, (s, Fx)
    assert s["fobj"] <= s["F"]                     # F: the master's running reference value (DPGOStar.cpp:200-205)


def test_conditioning_warning(fixtures_dir, capfd):
    """The factor holds explicit inverses of its pivot blocks: a pivot range beyond 1e13 is reported at
    construction; the BASELINE datasets stay silent."""
    n = 12
    I, J = np.arange(n - 1), np.arange(1, n)
    R = np.tile(np.eye(3), (n - 1, 1, 1))
    t = np.tile(np.array([1.0, 0.0, 0.0]), (n - 1, 1))
    kappa = np.full(n - 1, 100.0)
    tau = np.full(n - 1, 1e3)
    tau[2] = 3e-12                                # poses 0..2 hang on the rest by a measurement 15 orders weaker
    G = dpgo_amd.graph_from_edges(3, n, I, J, R, t, kappa, tau, 2)
    capfd.readouterr()
    dpgo_amd.NodeGroup(G, [0, 1], dpgo_amd.Options.driver(LOSS_NONE, True))
    assert "badly conditioned" in capfd.readouterr().err
    G2 = dpgo_amd.read_g2o(os.path.join(fixtures_dir, "sphere2500.g2o"), 1)   # one node: G_tt = Laplacian + 1e-11 I
    dpgo_amd.NodeGroup(G2, [0], dpgo_amd.Options.driver(LOSS_NONE, True))
    assert "WARNING" not in capfd.readouterr().err


def test_verbose_option_prints_and_changes_nothing(fixtures_dir, capfd):
    """Options::verbose (DPGO_types.h:87): one line per node and refinement, same iterates."""
    G = dpgo_amd.read_g2o(os.path.join(fixtures_dir, "smallGrid3D.g2o"), 2)
    outs = []
    for verbose in (0, 1):
        drv = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(LOSS_HUBER, True, verbose=verbose))
        capfd.readouterr()
        for _ in range(5):
            assert drv.step() == 0
        text = capfd.readouterr().out
        assert ("TNT:" in text) == bool(verbose)
        outs.append(drv.X().copy())
    assert np.array_equal(outs[0], outs[1])


def test_deferred_update_readback_changes_nothing(fixtures_dir, tmp_path):
    """update() returns before its scalars are read back (Group::finish_update) unless DPGO_DEFER_UPDATE=0; both ways give
    the same bits, through refinements, restarts and a per-node calling pattern (two processes: the switch is read once)."""
    import subprocess
    import sys
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import dpgo_amd
G = dpgo_amd.read_g2o(%r, 3)
drv = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(1, True))
trace = []
for it in range(30):
    assert drv.step() == 0
    trace.append([drv.group.results(a).fobj for a in range(3)] + [drv.group.results(a).gamma for a in range(3)])
grp = drv.group
for it in range(6):                       # nodes one by one, results read in between
    for a in range(3):
        assert grp.iterate([a]) == 0
    assert grp.communicate_local() == 0
    for a in (2, 0, 1):
        assert grp.update([a]) == 0
        trace.append([grp.results(a).fobj, grp.results(a).Gk, 0, 0, 0, 0])
np.savez(sys.argv[1], X=drv.X(), trace=np.array(trace))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(fixtures_dir, "torus3D.g2o"))
    outs = []
    for v in ("0", "1"):
        path = str(tmp_path / ("defer%s.npz" % v))
        subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, DPGO_DEFER_UPDATE=v))
        outs.append(np.load(path))
    assert np.array_equal(outs[0]["X"], outs[1]["X"])
    assert np.array_equal(outs[0]["trace"], outs[1]["trace"])


def test_engineering_switches_do_not_change_the_results(fixtures_dir, tmp_path):
    """The switches of DESIGN 8 select HOW something is computed, never WHAT: round 5's launch sequence (DPGO_FUSED=0: the
    extrapolation, the proximal step, Dfobj, iterate()'s tail, the retraction and the start of a refinement as launches of their
    own) is bit for bit the same run as round 6's fused one; the host numeric factorisation (DPGO_SPD_HOST_FACTOR=1), panels packed on the host
    (DPGO_SPD_DEVICE_PANELS=0) and the scalar-graph ordering (DPGO_SPD_QUOTIENT=0) are other exact factorisations of the
    same matrices, so the iterates agree to rounding (1e-9 after 25 iterations with refinements)."""
    import subprocess
    import sys
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import dpgo_amd
G = dpgo_amd.read_g2o(%r, 4)
drv = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(1, True))
for it in range(25):
    assert drv.step() == 0
np.save(sys.argv[1], drv.X())
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(fixtures_dir, "sphere2500.g2o"))

    def run(tag, **env):
        path = str(tmp_path / (tag + ".npy"))
        subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, **env))
        return np.load(path)

    base = run("base")
    # round 6: the fused launch sequence against round 5's (eager and replayed)
    assert np.array_equal(run("unfused", DPGO_FUSED="0"), base)
    assert np.array_equal(run("unfused_graph", DPGO_FUSED="0", DPGO_ITER_GRAPH="1"), base)
    # ... update()'s launches enqueued ahead of the host's acceptance decision, under the device-side gate (Group::SpecUpdate),
    # against the host deciding first
    assert np.array_equal(run("no_spec_update", DPGO_SPEC_UPDATE="0"), base)
    # round 5: the branch-free segments of an iteration (update() behind the exchange, the head of iterate(), a refinement
    # up to its trial point, every further CG step) replayed from captured HIP graphs -- the default for a group this
    # small -- against eager launches; and just the CG steps eager (round 4's switch)
    assert np.array_equal(run("nograph", DPGO_ITER_GRAPH="0"), base)
    assert np.array_equal(run("graph_forced", DPGO_ITER_GRAPH="1"), base)
    assert np.array_equal(run("no_cg_graph", DPGO_CG_GRAPH="0"), base)
    # ... the refinement started ahead of update()'s read-back (and, replayed, in one segment with the head of the
    # iteration) against the refinement that waits for the decision: the same launches in the same order
    assert np.array_equal(run("no_spec_refine", DPGO_SPEC_REFINE="0"), base)
    assert np.array_equal(run("no_spec_refine_graph", DPGO_SPEC_REFINE="0", DPGO_ITER_GRAPH="1"), base)
    # ... the factorisation's assembly with a launch per child slot against all children of a level gathered by the
    # parents' rows in one launch: the same additions in the same order
    assert np.array_equal(run("extend_by_slots", DPGO_SPD_EXTEND_SLOTS="1"), base)
    for tag, env in (("hostfactor", dict(DPGO_SPD_HOST_FACTOR="1")), ("hostpanels", dict(DPGO_SPD_DEVICE_PANELS="0")),
                     ("scalarorder", dict(DPGO_SPD_QUOTIENT="0")),
                     # round 3: the tree roots in two sweeps
                     ("tworootsweeps", dict(DPGO_SPD_FUSE_ROOT="0")),
                     # round 4: the fused roots stored as one triangle (forced on: these roots are below its size threshold)
                     ("roots_one_triangle_2_blocks", dict(DPGO_SPD_ROOT_SYM="1", DPGO_SPD_ROOT_SYM_BLOCKS="2")),
                     ("roots_one_triangle_8_blocks", dict(DPGO_SPD_ROOT_SYM="1", DPGO_SPD_ROOT_SYM_BLOCKS="8")),
                     # the factorisation's block columns right-looking at every level (another order of the same updates)
                     ("rightlooking", dict(DPGO_SPD_LEFT_LOOKING="0")),
                     # ... with the diagonal blocks in a launch of their own (the path of levels with thousands of fronts)
                     ("rightlooking_unfused", dict(DPGO_SPD_LEFT_LOOKING="0", DPGO_SPD_FUSE_POTRF_WGS="0"))):
        np.testing.assert_allclose(run(tag, **env), base, rtol=0, atol=1e-9, err_msg=tag)
    # Rescale::Dynamic: the device path (decision, block-diagonal rebuild, refactorisation from device-resident values)
    # against the host path (weights read back, nodes re-assembled, operators re-uploaded)
    dyn = code.replace("Options.driver(1, True)", "Options.driver(1, True, rescale=1)")
    assert dyn != code

    def run_dyn(tag, **env):
        path = str(tmp_path / (tag + ".npy"))
        subprocess.check_call([sys.executable, "-c", dyn, path], env=dict(os.environ, **env))
        return np.load(path)

    np.testing.assert_allclose(run_dyn("dyn_host", DPGO_RESCALE_HOST="1"), run_dyn("dyn_dev"), rtol=0, atol=1e-9, err_msg="dynamic")


def test_failed_refactorisation_after_a_rescale_fails_the_group(fixtures_dir, tmp_path):
    """Rescale::Dynamic on the device commits its scales before the verdict on the re-factored G_tt is read (the host never
    waits for it).  A non-positive pivot there -- which the reference reports from inside its Cholesky call
    (DPGOProblem.cpp:315, 479) -- must not leave a half-updated group computing on a broken factor: the call that reads
    the verdict returns -1 and so does every update() / iterate() after it.  (The verdict is forced by a test hook.)"""
    import subprocess
    import sys
    code = """
import sys
sys.path.insert(0, %r)
import dpgo_amd
G = dpgo_amd.read_g2o(%r, 2)
drv = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(1, True, rescale=1))
rcs = [drv.step() for _ in range(12)]
first = next(i for i, r in enumerate(rcs) if r != 0)
assert all(r != 0 for r in rcs[first:]), rcs
assert drv.group.iterate() == -1 and drv.group.update() == -1
print("failed at step", first)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(fixtures_dir, "smallGrid3D.g2o"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DPGO_DEBUG_FAIL_REFACTOR="1"), capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "failed at step" in out.stdout and "not positive definite after a rescale" in out.stderr


def test_update_goes_out_ahead_of_the_hosts_decision_and_falls_back(tmp_path):
    """Round 6 (Group::SpecUpdate): in the early regime of a robust, statically scaled run driven through step(), update()'s
    launches are enqueued ahead of the host's acceptance decision under the device-side gate; where the iteration leaves the
    common course (a node needs a second CG step, a restart, ...) the gated launches fall through and the host path runs.
    The lattice run below does both (the counters of DPGO_HOST_TIMING=1 say so), and its trajectory is bit for bit the one
    of the run in which the host always decides first."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import dpgo_amd
from dpgo_amd import synthetic
g = synthetic.grid(20, 20, 16, 25600)
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
drv = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(1, True), X0=G.chordal_initialization())
tr = []
for it in range(90):
    assert drv.step() == 0
    tr.append([drv.group.results(a).fobj for a in range(8)] if it %% 10 == 9 else [0.0] * 8)
np.savez(sys.argv[1], X=drv.X(), tr=np.array(tr))
del drv
""" % root

    def run(tag, **env):
        path = str(tmp_path / (tag + ".npz"))
        p = subprocess.run([sys.executable, "-c", code, path], env=dict(dict(os.environ, DPGO_HOST_TIMING="1", DPGO_ITER_GRAPH="0"), **env),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [l for l in p.stderr.splitlines() if "updates enqueued ahead" in l]
        assert line, p.stderr[-2000:]
        nums = [int(x) for x in line[-1].replace(",", " ").replace(":", " ").split() if x.isdigit()]
        return np.load(path), nums[-2], nums[-1]

    spec, enq, stood = run("spec")
    host, enq0, stood0 = run("host", DPGO_SPEC_UPDATE="0")
    assert enq0 == 0 and stood0 == 0
    assert enq >= 10 and 0 < stood <= enq, (enq, stood)
    assert np.array_equal(spec["X"], host["X"]) and np.array_equal(spec["tr"], host["tr"])
    # The results do not depend on how far the stream runs ahead of the host, nor on whether the CG steps go out as graph
    # replays (whose launches cover every node) or eagerly (sized for the live ones).  The multi-step CG of this run has few
    # live nodes at its late steps, where the fused roots of the solves take a finer tile class with a different summation
    # order: the class must follow from the nodes live after the step the host WAITED for (the summary carries each node's
    # stop ordinal), not from whatever later summary a late host happens to read (rounds <= 5 did that: a run with a cold
    # host parted from a warm one at iteration ~26), and a captured step must take the eager step's class.
    late, _, _ = run("late", DPGO_DEBUG_LATE_HOST_US="200")
    assert np.array_equal(spec["X"], late["X"]) and np.array_equal(spec["tr"], late["tr"])
    replay, _, _ = run("replay", DPGO_ITER_GRAPH="1")
    assert np.array_equal(spec["X"], replay["X"]) and np.array_equal(spec["tr"], replay["tr"])
    replay_late, _, _ = run("replay_late", DPGO_ITER_GRAPH="1", DPGO_DEBUG_LATE_HOST_US="60")
    assert np.array_equal(spec["X"], replay_late["X"])
