#!/usr/bin/env python3
"""Bit-level A/B of two builds of the library (DPGO_AMD_LIB) or of two settings of a switch: N iterations of a headline-like
lattice (three option sets) and of the parity datasets (robust and trivial loss, SE(3) and SE(2), one and several nodes,
AMM-PGO# and AMM-PGO*, long enough to leave the early regime), X and the per-node scalars dumped to a file.
Usage: ab_bits.py out.npz          then          ab_bits.py --compare a.npz b.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint64), b[k].view(np.uint64))
        if not same:
            bad += 1
            d = np.abs(a[k] - b[k]).max() if a[k].shape == b[k].shape else float("nan")
            print("DIFF %-24s max |a - b| = %.3e" % (k, d))
    print("%d arrays, %d differ: %s" % (len(a.files), bad, "BITWISE EQUAL" if bad == 0 else "NOT EQUAL"))
    sys.exit(1 if bad else 0)

import dpgo_amd
from dpgo_amd import synthetic
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FIX = os.path.join(ROOT, "fixtures", "g2o")
out = {}


SCALE = float(os.environ.get("AB_ITERS_SCALE", "1"))   # (longer runs: deeper into the interior regime)


def run(tag, G, opt, iters, X0=None):
    iters = int(iters * SCALE)
    drv = dpgo_amd.DistPGO(G, opt, X0=G.chordal_initialization() if X0 is None else X0)
    n = G.num_nodes
    tr = []
    for it in range(iters):
        assert drv.step() == 0
        r = [drv.group.results(a) for a in range(n)]
        tr.append([x.fobj for x in r] + [x.f for x in r] + [x.Gk for x in r] + [x.gradFnorm for x in r] + [float(x.tnt_inner_iterations) for x in r])
    out["X_" + tag] = drv.X()
    out["tr_" + tag] = np.array(tr)


g = synthetic.grid(20, 20, 16, 25600)
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
for tag, opt in (("huber", dpgo_amd.Options.driver(1, True)), ("gm_mm", dpgo_amd.Options.driver(2, False)), ("dyn", dpgo_amd.Options.driver(1, True, rescale=1))):
    run("lattice_" + tag, G, opt, 25)
for name, nn, loss, acc, iters in (("sphere2500", 1, 1, True, 60), ("torus3D", 8, 1, True, 80), ("city10000", 8, 1, True, 60),
                                   ("smallGrid3D", 2, 0, False, 40), ("torus3D", 4, 0, True, 40), ("M3500", 4, 3, True, 40)):
    Gd = dpgo_amd.read_g2o(os.path.join(FIX, name + ".g2o"), nn)
    run("%s_%d_%d_%d" % (name, nn, loss, int(acc)), Gd, dpgo_amd.Options.driver(loss, acc), iters)
# AMM-PGO* (the master's sums: DPGOStar)
Gs = dpgo_amd.read_g2o(os.path.join(FIX, "M3500.g2o"), 4)
st = dpgo_amd.DPGOStar(Gs, dpgo_amd.Options.driver(1, True))
assert st.initialize(Gs.chordal_initialization()) == 0
trs = []
for it in range(int(30 * SCALE)):
    assert st.step() == 0
    s_ = st.state()
    trs.append([st.group.results(a).fobj for a in range(4)] + [s_["F"], s_["fobj"], s_["fobjh"], float(s_["branches"])])
out["tr_star"] = np.array(trs)
out["X_star"] = st.X()
np.savez(sys.argv[1], **out)
