#!/usr/bin/env python3
"""Generate tests/golden/oracle_traces.json: per-iteration (2F, 2|grad F|) traces of the oracle's dist_pgo
driver loop (oracle/star.py DistPGO, chordal initialisation) on the BASELINE.json parity configurations.

The reference itself cannot be built here, so these traces do NOT pin the oracle
against the reference; they freeze the oracle's behaviour (a change in oracle/ that moves a trace is
caught on CPU) and give the GPU tests fixed numbers to hit at the full iteration counts.
Usage: python tests/golden/make_oracle_traces.py   (a few minutes of CPU)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import g2o as og                                   # noqa: E402
from oracle.hash import Options                                 # noqa: E402
from oracle.problem import LOSS_HUBER, LOSS_NONE, LOSS_WELSCH   # noqa: E402
from oracle.star import DistPGO, chordal_initialization         # noqa: E402

# name, dataset, nodes, loss, accelerated, iterations
CASES = [
    ("config1_smallGrid3D_mm_2nodes", "smallGrid3D", 2, LOSS_NONE, False, 200),     # BASELINE configs[0]
    ("config2_sphere2500_amm_1node", "sphere2500", 1, LOSS_NONE, True, 300),        # configs[1], first 300 of 1000
    ("config3_torus3D_amm_8nodes", "torus3D", 8, LOSS_NONE, True, 60),              # configs[2]
    ("config3_city10000_amm_8nodes", "city10000", 8, LOSS_NONE, True, 40),          # configs[2], SE(2)
    ("tinyGrid3D_amm_huber_2nodes", "tinyGrid3D", 2, LOSS_HUBER, True, 100),
    ("smallGrid3D_amm_welsch_4nodes", "smallGrid3D", 4, LOSS_WELSCH, True, 100),
    ("M3500_amm_4nodes", "M3500", 4, LOSS_NONE, True, 40),                          # SE(2)
]


def main():
    out = {"generator": "tests/golden/make_oracle_traces.py", "cases": {}}
    for name, ds, nn, loss, acc, iters in CASES:
        path = os.path.join(ROOT, "fixtures", "g2o", ds + ".g2o")
        t0 = time.time()
        num_poses, mm = og.read_g2o_file(path)
        X0 = chordal_initialization(num_poses, mm)
        drv = DistPGO(path, nn, Options.driver(loss, acc), X0=X0, mm=mm, num_poses=num_poses)
        trace = [list(drv.evaluate())]
        for _ in range(iters):
            drv.step(evaluate=False)
            trace.append(list(drv.evaluate()))
        out["cases"][name] = {"dataset": ds, "num_nodes": nn, "loss": int(loss), "accelerated": bool(acc),
                              "iterations": iters, "trace_2F_2gradnorm": trace}
        print("%-34s %5.1f s   2F: %.10g -> %.10g" % (name, time.time() - t0, trace[0][0], trace[-1][0]), flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "oracle_traces.json"), "w") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    main()
