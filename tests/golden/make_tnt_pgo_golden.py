"""Golden traces of the REFERENCE's TNT / STPCG on DPGO-shaped problems.

Runs only in the build container (needs /root/reference for oracle/_ref/tnt_pgo_ref, built by `make -C oracle/ref_tnt`).
For every case below the oracle is driven to the point where DPGOHash calls TNT (DPGOHash.cpp:270-349); the node's
surrogate (dense G, g, f, G_tt^-1, (G_RR + lambda I)^-1), the start point and the solver parameters are written to a
problem file; oracle/ref_tnt/harness_pgo.cpp runs Optimization::Riemannian::TNT (the reference's own header) on it with
its own dense restatement of the operators; the solver's trace and result are committed as tests/golden/tnt_pgo_ref.jsonl
together with the recipe (dataset, options, node, call number) that lets the tests rebuild the same problem.

  python tests/golden/make_tnt_pgo_golden.py
"""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = [
    # name, dataset, num_nodes, loss, accelerated, node, which TNT call of that node (0 = first), option overrides
    ("smallGrid3D_mm_node0_call0", "smallGrid3D", 2, "trivial", False, 0, 0, {}),
    ("smallGrid3D_mm_node1_call2", "smallGrid3D", 2, "trivial", False, 1, 2, {}),
    ("smallGrid3D_amm_node0_call1", "smallGrid3D", 2, "trivial", True, 0, 1, {}),
    ("smallGrid3D_huber_mm_node0_call0", "smallGrid3D", 2, "huber", False, 0, 0, {}),
    ("smallGrid3D_mm_3accepted", "smallGrid3D", 2, "trivial", False, 0, 0, {"max_iterations_accepted": 3}),
    ("smallGrid3D_mm_tight", "smallGrid3D", 2, "trivial", False, 1, 0,
     {"max_iterations_accepted": 10, "max_iterations": 20, "rel_func_decrease_tol": 1e-12, "stepsize_tol": 1e-9}),
    ("tinyGrid3D_mm_node0_call0", "tinyGrid3D", 2, "trivial", False, 0, 0, {}),
]


def capture(dataset, num_nodes, loss, accelerated, node, call, overrides, max_outer=30):
    """Drive the oracle until node `node` makes its `call`-th TNT call; return what it was called with."""
    from oracle import g2o as og
    from oracle import hash as H
    from oracle.hash import Options
    from oracle.problem import LOSS_NAMES
    from oracle.star import DistPGO, chordal_initialization
    path = os.path.join(ROOT, "fixtures", "g2o", dataset + ".g2o")
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    opt = Options.driver(LOSS_NAMES[loss], accelerated)
    for k, v in overrides.items():
        setattr(opt, k, v)
    dp = DistPGO(path, num_nodes, opt, X0=X0, mm=mm, num_poses=num_poses)
    got = {}
    count = [0]
    real_tnt = H.tnt_mod.tnt
    nd = dp.nodes[node]

    def spy(Fobj, QM, metric, retract, x0, precon, prm, log=None):
        mine = (Fobj.__closure__ is not None) and any(c.cell_contents is nd.problem for c in Fobj.__closure__ if hasattr(c, "cell_contents"))
        lg = []
        res = real_tnt(Fobj, QM, metric, retract, x0, precon, prm, lg)
        if mine:
            if count[0] == call and not got:
                cells = {n: c.cell_contents for n, c in zip(Fobj.__code__.co_freevars, Fobj.__closure__)}
                got.update(x0=np.array(x0), g=np.array(cells["g"]), f=float(cells["f"]), prm=prm, res=res, log=lg, outer=dp.nodes[node].results.iters)
            count[0] += 1
        return res

    H.tnt_mod.tnt = spy
    try:
        for _ in range(max_outer):
            dp.step(evaluate=False)
            if got:
                break
    finally:
        H.tnt_mod.tnt = real_tnt
    if not got:
        raise RuntimeError("node %d never made TNT call %d" % (node, call))
    return dp, nd, got


def write_problem(path, nd, got):
    p = nd.problem
    d, n0 = p.d, p.n[0]
    G = np.asarray(p.mat.G.todense(), dtype=np.float64)
    Gtt = np.asarray(p.mat.Gtt.todense(), dtype=np.float64)
    GttInv = np.linalg.inv(Gtt)
    has_precon = p.precon is not None
    prm = got["prm"]
    pv = [prm.max_iterations, prm.max_iterations_accepted, prm.gradient_tolerance, prm.preconditioned_gradient_tolerance,
          prm.relative_decrease_tolerance, prm.stepsize_tolerance, prm.kappa_fgr, prm.theta, prm.max_TPCG_iterations]
    with open(path, "wb") as fh:
        fh.write(struct.pack("<iiii", d, n0, int(has_precon), len(pv)))
        fh.write(np.ascontiguousarray(G).tobytes())
        fh.write(np.ascontiguousarray(got["g"], dtype=np.float64).tobytes())
        fh.write(struct.pack("<d", got["f"]))
        fh.write(np.ascontiguousarray(GttInv).tobytes())
        if has_precon:
            M = np.asarray(p.mat.GRR.todense(), dtype=np.float64) + (p.lambda_max / 1e6) * np.eye(d * n0)
            fh.write(np.ascontiguousarray(np.linalg.inv(M)).tobytes())
        fh.write(np.ascontiguousarray(got["x0"], dtype=np.float64).tobytes())
        fh.write(np.asarray(pv, dtype=np.float64).tobytes())


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "tnt_pgo_ref")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "ref_tnt")])
    out_lines = []
    tmp = tempfile.mkdtemp(prefix="tnt_pgo_")
    for name, dataset, nn, loss, acc, node, call, ov in CASES:
        dp, nd, got = capture(dataset, nn, loss, acc, node, call, ov)
        prob = os.path.join(tmp, name + ".bin")
        write_problem(prob, nd, got)
        ref = json.loads(subprocess.check_output([exe, name, prob], text=True).strip().splitlines()[-1])
        ref["recipe"] = {"dataset": dataset, "num_nodes": nn, "loss": loss, "accelerated": acc, "node": node, "call": call,
                         "overrides": ov, "outer_iteration": got["outer"]}
        # how far the oracle is from the reference on this case, for the record (the test asserts it)
        o = got["res"]
        print("%-36s ref f %.15g status %d inner %s | oracle f %.15g inner %s | |dx| %.2e" % (
            name, ref["f"], ref["status"], [int(v) for v in ref["inner_iterations"]], o["f"], [l["inner"] for l in got["log"]],
            float(np.abs(np.asarray(ref["x"]).reshape(o["x"].shape) - o["x"]).max())))
        out_lines.append(json.dumps(ref))
        os.remove(prob)
    os.rmdir(tmp)
    with open(os.path.join(ROOT, "tests", "golden", "tnt_pgo_ref.jsonl"), "w") as fh:
        fh.write("\n".join(out_lines) + "\n")
    print("wrote tests/golden/tnt_pgo_ref.jsonl (%d cases, %d bytes)" % (len(out_lines), sum(len(l) for l in out_lines)))


if __name__ == "__main__":
    main()
