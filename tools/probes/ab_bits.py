#!/usr/bin/env python3
"""Bit-level A/B of two builds of the library (DPGO_AMD_LIB): N iterations of the headline-like lattice and of a dataset,
X and the per-node scalars dumped to a file.  Usage: ab_bits.py out.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import dpgo_amd
from dpgo_amd import synthetic
out = {}
g = synthetic.grid(20, 20, 16, 25600)
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
for tag, opt in (("huber", dpgo_amd.Options.driver(1, True)), ("gm_mm", dpgo_amd.Options.driver(2, False)), ("dyn", dpgo_amd.Options.driver(1, True, rescale=1))):
    drv = dpgo_amd.DistPGO(G, opt, X0=G.chordal_initialization())
    tr = []
    for it in range(25):
        assert drv.step() == 0
        tr.append([drv.group.results(a).fobj for a in range(8)] + [drv.group.results(a).f for a in range(8)])
    out["X_" + tag] = drv.X()
    out["tr_" + tag] = np.array(tr)
np.savez(sys.argv[1], **out)
