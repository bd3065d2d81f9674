#!/usr/bin/env python3
"""Iterations / s of one parity configuration over the SAME window tests/config_rates.py times (3 warm-up + n iterations from
the chordal initialisation: the phase with the long truncated-CG runs), under the current environment.
Usage: python tools/probes/config_phase.py <dataset> <nodes> <loss 0|1> <iters>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dpgo_amd
from oracle import g2o as og
from oracle.star import chordal_initialization
ds, nn, loss, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
path = os.path.join(ROOT, "fixtures", "g2o", ds + ".g2o")
num_poses, mm = og.read_g2o_file(path)
X0 = chordal_initialization(num_poses, mm)
G = dpgo_amd.read_g2o(path, nn)
best, inner = 0, 0
for rep in range(3):
    gpu = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(loss, True), X0=X0)
    for _ in range(3):
        gpu.step()
    gpu.group.sync()
    t0 = time.perf_counter()
    inner = 0
    for _ in range(iters):
        gpu.step()
        inner += max(int(gpu.group.results(k).tnt_inner_iterations) for k in range(nn))
    gpu.group.sync()
    best = max(best, iters / (time.perf_counter() - t0))
    del gpu
print("%s nodes %d loss %d: %.1f it/s (best of 3 x %d from the chordal initialisation); CG steps (max over nodes) per iteration %.1f  env %s" % (
    ds, nn, loss, best, iters, inner / iters, {k: v for k, v in os.environ.items() if k.startswith("DPGO_")}))
