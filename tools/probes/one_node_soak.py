"""One rank of an 8-GPU run (node 3 of the headline graph, frozen neighbours) over the whole refinement history: 260
iterations, objective trace and wall time -- run once with DPGO_CG_GRAPH=0 and once with 1 to compare (same bits expected)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, dpgo_amd
from dpgo_amd import synthetic
g = synthetic.grid(50, 50, 40, 400000, seed=synthetic.HEADLINE["seed"])
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
X0 = G.chordal_initialization()
node = int(sys.argv[2]) if len(sys.argv) > 2 else 3
grp = dpgo_amd.NodeGroup(G, [node], dpgo_amd.Options.driver(dpgo_amd.LOSS_HUBER, True))
grp.initialize_global(X0); grp.update(); grp.sync()
t0 = time.perf_counter(); tr = []; inner = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 260):
    assert grp.step(None) == 0
    r = grp.results(0)
    tr.append(r.fobj); inner += int(r.tnt_inner_iterations) if r.refined else 0
grp.sync()
dt = time.perf_counter() - t0
import hashlib
print("node %d " % node + "graph=%s: %d iterations %.3f s, %d CG steps, fobj[-1] = %.12e, trace md5 %s" % (os.environ.get("DPGO_CG_GRAPH", "auto"), len(tr), dt, inner, tr[-1],
      hashlib.md5(np.asarray(tr).tobytes()).hexdigest()))
