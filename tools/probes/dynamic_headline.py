#!/usr/bin/env python3
"""Rescale::Dynamic (the reference's default) at the headline size: iterations / s, rescale events, objective."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dpgo_amd
from dpgo_amd import synthetic
dims = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "50,50,40,400000").split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
g = synthetic.grid(*dims, seed=synthetic.HEADLINE["seed"])
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
X0 = G.chordal_initialization()
for rescale in ((1,) if os.environ.get("DYN_ONLY") else (0, 1)):
    opt = dpgo_amd.Options.driver(1, True, rescale=rescale)
    grp = dpgo_amd.NodeGroup(G, list(range(8)), opt)
    grp.initialize_global(X0)
    grp.update()
    t0 = time.perf_counter()
    trace = []
    for it in range(n):
        assert grp.iterate() == 0
        grp.communicate_local()
        assert grp.update() == 0
        if it < 12 or it == n - 1:
            trace.append("%d:%.6e" % (it, sum(grp.results(k).fobj for k in range(8))))
    grp.sync()
    dt = time.perf_counter() - t0
    print("rescale=%d: %d iterations in %.3f s (%.2f ms / iteration)" % (rescale, n, dt, dt / n * 1e3))
    print("   ", " ".join(trace))
    del grp
