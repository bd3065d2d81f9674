"""Headline group (8 nodes on one GPU) run to iteration n (default 140: the interior regime, long truncated CGs), so that the
tail of a kernel trace shows CG steps with few live nodes.  Usage under rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dpgo_amd
from dpgo_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 140
g = synthetic.grid(50, 50, 40, 400000, seed=synthetic.HEADLINE["seed"])
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
X0 = G.chordal_initialization()
grp = dpgo_amd.NodeGroup(G, list(range(8)), dpgo_amd.Options.driver(dpgo_amd.LOSS_HUBER, True))
grp.initialize_global(X0); grp.update()
for it in range(n):
    assert grp.step(None) == 0
grp.sync()
print([int(grp.results(k).tnt_inner_iterations) for k in range(8)])
