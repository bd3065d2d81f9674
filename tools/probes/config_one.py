#!/usr/bin/env python3
"""Iterations / s of ONE parity configuration under the current environment (for same-box A/B of switches).
Usage: [RESCALE=1] python tools/probes/config_one.py <dataset> <nodes> <loss 0|1> <iters>   (RESCALE=1: Rescale::Dynamic)"""
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
gpu = dpgo_amd.DistPGO(dpgo_amd.read_g2o(path, nn), dpgo_amd.Options.driver(loss, True, rescale=int(os.environ.get("RESCALE", "0"))), X0=X0)
for _ in range(3):
    gpu.step()
gpu.group.sync()
best = 0
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(iters):
        gpu.step()
    gpu.group.sync()
    best = max(best, iters / (time.perf_counter() - t0))
r = [gpu.group.results(k) for k in range(nn)]
print("%s nodes %d loss %d: %.1f it/s (best of 3 x %d)  refined %d inner %d  env %s" % (
    ds, nn, loss, best, iters, sum(int(x.refined) for x in r), sum(int(x.tnt_inner_iterations) for x in r),
    {k: v for k, v in os.environ.items() if k.startswith("DPGO_")}))
