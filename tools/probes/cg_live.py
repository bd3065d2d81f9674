"""Per outer iteration of the headline run: wall time, CG steps of every node (the lockstep CG runs max over the
nodes; late steps run on few nodes).  Usage: python tools/probes/cg_live.py [iterations] > gpurun_out/cg_live.txt"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, dpgo_amd
from dpgo_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 220
g = synthetic.grid(50, 50, 40, 400000, seed=synthetic.HEADLINE["seed"])
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
opt = dpgo_amd.Options.driver(dpgo_amd.LOSS_HUBER, True)
X0 = G.chordal_initialization()
grp = dpgo_amd.NodeGroup(G, list(range(8)), opt, device=0)
grp.initialize_global(X0); grp.update(); grp.sync()
t0 = time.perf_counter(); prev = t0
for it in range(n):
    grp.iterate(); grp.communicate_local(); grp.update(); grp.sync()
    now = time.perf_counter()
    r = [grp.results(k) for k in range(8)]
    inner = [int(x.tnt_inner_iterations) if x.refined else -1 for x in r]
    print(it, "%.3f" % ((now - prev) * 1e3), "%.4f" % (now - t0), "%.9e" % (2 * sum(x.fobj for x in r)), inner, sum(int(x.restarts) for x in r), flush=True)
    prev = now
