import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, dpgo_amd
from dpgo_amd import synthetic
g = synthetic.grid(50,50,40,400000, seed=synthetic.HEADLINE["seed"])
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
opt = dpgo_amd.Options.driver(dpgo_amd.LOSS_HUBER, True)
X0 = G.chordal_initialization()
grp = dpgo_amd.NodeGroup(G, list(range(8)), opt, device=0)
grp.initialize_global(X0); grp.update(); grp.sync()
prev=time.perf_counter()
for it in range(260):
    grp.iterate(); grp.communicate_local(); grp.update()
    now=time.perf_counter()
    r=[grp.results(k) for k in range(8)]
    if it<12 or it%20==0: print(it, "%.2f ms"%((now-prev)*1e3), "2F %.6e"%(2*sum(x.fobj for x in r)), "refined", sum(int(x.refined) for x in r), "inner", sum(int(x.tnt_inner_iterations) for x in r), "restarts", sum(int(x.restarts) for x in r))
    prev=time.perf_counter()
