#!/usr/bin/env python3
"""Beyond the headline size: a lattice given on the command line (default 80x80x60 = 384 000 poses, 1.5 M edges, 8
nodes), a few AMM-PGO# iterations, sum_a fobj^a against the independent global cost pass, iterations / s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import dpgo_amd
from dpgo_amd import synthetic
dims = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "80,80,60,1536000").split(",")]
t0 = time.time()
g = synthetic.grid(*dims)
N = g["num_poses"]
G = dpgo_amd.graph_from_edges(3, N, g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
X0 = G.chordal_initialization()
t1 = time.time()
grp = dpgo_amd.NodeGroup(G, range(8), dpgo_amd.Options.driver(1, True))
t2 = time.time()
grp.initialize_global(X0); grp.update()
for _ in range(3):
    grp.iterate(); grp.communicate_local(); grp.update()
grp.sync()
t3 = time.perf_counter()
n = 10
for _ in range(n):
    assert grp.iterate() == 0 and grp.communicate_local() == 0 and grp.update() == 0
grp.sync()
dt = (time.perf_counter() - t3) / n
X = np.zeros((4 * N, 3), order="F")
grp.scatter_global(X)
Fsum = sum(grp.results(k).fobj for k in range(8))
F, g2 = grp.evaluate(X)
print("%d poses, %d edges: graph + init %.1f s, group %.1f s, %.2f ms / iteration; sum fobj %.9e  F %.9e  rel diff %.1e" % (
    N, len(g["I"]), t1 - t0, t2 - t1, dt * 1e3, Fsum, F, abs(Fsum - F) / F))
