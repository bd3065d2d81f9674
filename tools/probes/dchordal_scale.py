#!/usr/bin/env python3
"""The distributed chordal initialisation (dist_pgo --dist_init true) on a lattice the oracle cannot afford: time,
stage objectives, and the quality of the result as a starting point (F of the result vs F of the centralised init)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import dpgo_amd
from dpgo_amd import synthetic
dims = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "32,32,24,98304").split(",")]
nn = int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = synthetic.grid(*dims)
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], nn)
grp = dpgo_amd.NodeGroup(G, range(nn), dpgo_amd.Options.driver(1, True))
t0 = time.time()
X, obj = grp.dist_chordal_initialization()
t1 = time.time()
X0 = G.chordal_initialization()
t2 = time.time()
F_dist, _ = grp.evaluate(X)
F_cent, _ = grp.evaluate(X0)
print("dist-init %.2f s (centralised %.2f s); stage objectives first / last: %s ... %s" % (t1 - t0, t2 - t1, obj[:3], obj[-3:]))
print("F(dist-init) = %.6e   F(centralised chordal) = %.6e   finite %s" % (F_dist, F_cent, np.isfinite(X).all()))
