"""AMM-PGO* on M3500 / 4 nodes: iterations / s over 40 iterations after 3 (the window of tests/config_rates.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dpgo_amd
from oracle import g2o as og
from oracle.star import chordal_initialization
path = os.path.join(ROOT, "fixtures", "g2o", "M3500.g2o")
num_poses, mm = og.read_g2o_file(path)
X0 = chordal_initialization(num_poses, mm)
gpu = dpgo_amd.DPGOStar(dpgo_amd.read_g2o(path, 4), dpgo_amd.Options.driver(0, True))
gpu.initialize(X0)
for _ in range(3):
    gpu.step()
gpu.group.sync()
t0 = time.perf_counter()
inner = 0
for _ in range(40):
    gpu.step()
    inner += max(int(gpu.group.results(k).tnt_inner_iterations) for k in range(4))
gpu.group.sync()
print("M3500 star %.1f it/s, CG steps (max over nodes) per iteration %.1f, branches %s" % (40 / (time.perf_counter() - t0), inner / 40, gpu.state()))
