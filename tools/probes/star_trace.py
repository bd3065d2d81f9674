"""AMM-PGO* with Dynamic rescale on smallGrid3D / 2 nodes: fobj and the branches taken per iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, dpgo_amd
from oracle import g2o as og
from oracle.star import chordal_initialization
path = os.path.join(ROOT, "fixtures", "g2o", "smallGrid3D.g2o")
num_poses, mm = og.read_g2o_file(path)
X0 = chordal_initialization(num_poses, mm)
star = dpgo_amd.DPGOStar(dpgo_amd.read_g2o(path, 2), dpgo_amd.Options.driver(1, True, rescale=1))
star.initialize(X0)
for it in range(20):
    star.step()
    print(it, repr(star.state()))
