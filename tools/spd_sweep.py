#!/usr/bin/env python3
"""Solve-kernel sweep on the GPU box: for every value of an environment knob (DPGO_SPD_*), create the headline group
(8 nodes on one GPU, or --one: node 3 alone) with DPGO_SPD_DUMP=1 and print the per-factor totals the dump ends with
(MB, us, GB/s of one solve, launches timed one by one).  Usage: spd_sweep.py VAR v1 v2 ... [--one] [--levels]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
one, levels = "--one" in sys.argv, "--levels" in sys.argv
var, vals = args[0], args[1:]
code = """
import sys; sys.path.insert(0, %r)
import dpgo_amd
from dpgo_amd import synthetic
g = synthetic.grid(50, 50, 40, 400000, seed=synthetic.HEADLINE["seed"])
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
grp = dpgo_amd.NodeGroup(G, %s, dpgo_amd.Options.driver(1, True))
""" % (ROOT, "[3]" if one else "range(8)")
for v in vals:
    env = dict(os.environ, DPGO_SPD_DUMP="1")
    env[var] = v
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stderr
    print("== %s=%s" % (var, v))
    for line in out.splitlines():
        if (line.startswith("[spd]") and ("total" in line or (levels and "level" in line))) or (levels and line.startswith("[trace]")):
            print(line)
    sys.stdout.flush()
