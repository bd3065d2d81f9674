"""Run-to-run reproducibility probe: 8 nodes over a 25 600-edge lattice, Huber, 90 iterations; per iteration the
objective, gradient norm, inner-iteration and restart counts of every node and a checksum of its X block."""
import sys, os, zlib, numpy as np
sys.path.insert(0, os.getcwd())
import dpgo_amd
from dpgo_amd import synthetic
g = synthetic.grid(20, 20, 16, 25600)
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
drv = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(1, True), X0=G.chordal_initialization())
tr = []
for it in range(int(os.environ.get("STEPS", "90"))):
    assert drv.step() == 0
    r = [drv.group.results(a) for a in range(8)]
    tr.append([x.fobj for x in r] + [x.Gk for x in r] + [float(x.tnt_inner_iterations) for x in r]
              + [float(x.restarts) for x in r] + [float(zlib.crc32(c.tobytes())) for c in np.array_split(drv.X(), 8)])
np.save(sys.argv[1], np.array(tr))
