#!/usr/bin/env python3
"""Writes tests/golden/dchordal_oracle.json: the distributed chordal initialisation of smallGrid3D / 2 nodes and
M3500 / 4 nodes as computed by oracle/dchordal.py (stage objectives sampled every 20 iterations, checksum of the
result).  These are the ORACLE'S OWN outputs (the reference's DChordal cannot be built here: Eigen / CHOLMOD /
SE-Sync), kept to notice unintended changes of the restatement -- not reference-pinned vectors."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import g2o as og                      # noqa: E402
from oracle.dchordal import dist_chordal_initialization   # noqa: E402

out = {"note": __doc__.strip(), "cases": {}}
for name, nn in (("smallGrid3D", 2), ("M3500", 4)):
    num_poses, mm = og.read_g2o_file(os.path.join(ROOT, "fixtures", "g2o", name + ".g2o"))
    _, meas, _ = og.partition_measurements(num_poses, mm, nn)
    tr = {}
    Xk = dist_chordal_initialization(meas, trace=tr)
    out["cases"]["%s_%d" % (name, nn)] = {
        "objectives": {k: tr[k] for k in ("objective_reduced_R", "objective_R", "objective_reduced_t", "objective_t")},
        "sum_abs_X": [float(np.abs(X).sum()) for X in Xk],
        "first_rows": [X[:2].tolist() for X in Xk],
    }
with open(os.path.join(ROOT, "tests", "golden", "dchordal_oracle.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print("written")
