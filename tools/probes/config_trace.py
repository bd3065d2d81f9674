#!/usr/bin/env python3
"""One parity configuration, iteration by iteration: objective, nodes that refined, CG steps -- to compare two builds
(DPGO_AMD_LIB) on the same box.  Usage: python tools/probes/config_trace.py <dataset> <nodes> <loss 0|1> <iters> [every]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dpgo_amd
from oracle import g2o as og
from oracle.star import chordal_initialization
ds, nn, loss, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
every = int(sys.argv[5]) if len(sys.argv) > 5 else 10
path = os.path.join(ROOT, "fixtures", "g2o", ds + ".g2o")
num_poses, mm = og.read_g2o_file(path)
X0 = chordal_initialization(num_poses, mm)
gpu = dpgo_amd.DistPGO(dpgo_amd.read_g2o(path, nn), dpgo_amd.Options.driver(loss, True), X0=X0)
inner = refined = 0
t0 = time.perf_counter()
for it in range(iters):
    gpu.step()
    r = [gpu.group.results(k) for k in range(nn)]
    inner += sum(int(x.tnt_inner_iterations) for x in r)
    refined += sum(int(x.refined) for x in r)
    if it % every == every - 1:
        print("it %4d  sum fobj %.12e  refined so far %5d  CG steps so far %6d  %.1f ms" % (
            it + 1, sum(x.fobj for x in r), refined, inner, 1e3 * (time.perf_counter() - t0)))
