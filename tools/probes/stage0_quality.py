"""Row f2 evidence: how converged is the stage-0 stand-in of the distributed initialisation (dchordal.cpp: chordal
initialisation of a node's own subgraph + local_iters MM-PGO iterations with the refinement forced on; the reference
runs a per-node SE-Sync solve there, dist_pgo.cpp:146-158)?  Per node: objective and Riemannian gradient norm of the
node's LOCAL problem after local_iters and after 3 x local_iters iterations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import dpgo_amd


def local_problem(G, nn):
    """The graph without its inter-node edges (same contiguous partition) and the per-node chordal initial guess."""
    d, N = G.d, G.num_poses
    I, J, R, t, kappa, tau = G.edges()
    off = np.array([G.node_offset(a) for a in range(nn)] + [N])
    node = lambda p: np.searchsorted(off, p, side="right") - 1
    keep = node(I) == node(J)
    Gi = dpgo_amd.graph_from_edges(d, N, I[keep], J[keep], R[keep], t[keep], kappa[keep], tau[keep], nn)
    X = np.zeros(((d + 1) * N, d), order="F")
    for a in range(nn):
        m = keep & (node(I) == a)
        n0 = off[a + 1] - off[a]
        Ga = dpgo_amd.graph_from_edges(d, n0, I[m] - off[a], J[m] - off[a], R[m], t[m], kappa[m], tau[m], 1)
        Xa = Ga.chordal_initialization()
        X[off[a]:off[a + 1]] = Xa[:n0]
        X[N + off[a] * d:N + off[a + 1] * d] = Xa[n0:]
    return Gi, X


def run(path, nn, iters):
    G = dpgo_amd.read_g2o(path, nn)
    Gi, X = local_problem(G, nn)
    # the stand-in's options (dchordal.cpp): MM-PGO, trivial loss, refinement in every iteration, RegularizedCholesky
    opt = dpgo_amd.Options.driver(dpgo_amd.LOSS_NONE, False, accepted_delta=0.0, preconditioner=3)
    grp = dpgo_amd.NodeGroup(Gi, range(nn), opt)
    assert grp.initialize_global(X) == 0 and grp.update() == 0
    out = {}
    for it in range(1, max(iters) + 1):
        assert grp.iterate() == 0 and grp.update() == 0
        if it in iters:
            out[it] = [(grp.results(k).fobj, grp.results(k).gradFnorm) for k in range(nn)]
    return out


if __name__ == "__main__":
    L = dpgo_amd.DChordalOptions().local_iters
    for name, nn in (("smallGrid3D", 2), ("tinyGrid3D", 2), ("sphere2500", 4), ("torus3D", 8), ("M3500", 4), ("city10000", 8)):
        r = run(os.path.join(ROOT, "fixtures", "g2o", name + ".g2o"), nn, (L, 3 * L))
        for k in range(nn):
            (f1, g1), (f3, g3) = r[L][k], r[3 * L][k]
            print("%-12s node %d: after %3d it F = %.10e |grad| = %.3e ; after %3d it F = %.10e |grad| = %.3e ; rel dF = %.2e"
                  % (name, k, L, f1, g1, 3 * L, f3, g3, (f1 - f3) / max(abs(f3), 1e-300)))
