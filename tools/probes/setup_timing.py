#!/usr/bin/env python3
"""Where the set-up time of the headline group goes (DPGO_SETUP_TIMING=1 prints the phases on stderr)."""
import os, sys, time
os.environ["DPGO_SETUP_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dpgo_amd
from dpgo_amd import synthetic
t0 = time.time()
g = synthetic.grid(50, 50, 40, 400000, seed=synthetic.HEADLINE["seed"])
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
print("graph %.2f s" % (time.time() - t0)); t0 = time.time()
X0 = G.chordal_initialization()
print("chordal init %.2f s" % (time.time() - t0)); t0 = time.time()
grp = dpgo_amd.NodeGroup(G, list(range(8)), dpgo_amd.Options.driver(1, True))
print("group %.2f s" % (time.time() - t0))
