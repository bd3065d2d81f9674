"""Long runs: N iterations of a lattice (default 20x20x16 / 25 600 edges / 8 nodes, Huber; --headline: 50x50x40 / 400 000) --
the run must stay finite, end at the same bits as a second run of the same settings, and print its objective every 100
iterations.  Usage: soak.py out.npz [iterations] [--headline] [--dynamic]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import dpgo_amd
from dpgo_amd import synthetic
args = [a for a in sys.argv[1:] if not a.startswith("--")]
iters = int(args[1]) if len(args) > 1 else 1000
dims = (50, 50, 40, 400000) if "--headline" in sys.argv else (20, 20, 16, 25600)
g = synthetic.grid(*dims)
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
opt = dpgo_amd.Options.driver(1, True, rescale=1) if "--dynamic" in sys.argv else dpgo_amd.Options.driver(1, True)
drv = dpgo_amd.DistPGO(G, opt, X0=G.chordal_initialization())
t0 = time.time()
tr = []
for it in range(iters):
    assert drv.step() == 0, it
    if it % 100 == 99 or it == iters - 1:
        F2, g2 = drv.evaluate()
        assert np.isfinite(F2) and np.isfinite(g2), (it, F2, g2)
        tr.append((it + 1, F2, g2))
        print("%5d  2F = %.10e  2|grad| = %.3e  %.1f s" % (it + 1, F2, g2, time.time() - t0), flush=True)
X = drv.X()
assert np.isfinite(X).all()
np.savez(args[0], X=X, tr=np.array(tr))
