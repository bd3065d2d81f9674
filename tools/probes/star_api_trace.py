"""AMM-PGO* on M3500 / 4 nodes THROUGH an RCCL communicator of one rank (the master's sums go k_star_sums -> ncclAllReduce on
the group's stream -> k_publish): 3 + 10 iterations, for a kernel + HIP API trace (tools/probes/star_api_summary.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dpgo_amd
from oracle import g2o as og
from oracle.star import chordal_initialization
path = os.path.join(ROOT, "fixtures", "g2o", "M3500.g2o")
num_poses, mm = og.read_g2o_file(path)
X0 = chordal_initialization(num_poses, mm)
star = dpgo_amd.DPGOStar(dpgo_amd.read_g2o(path, 4), dpgo_amd.Options.driver(0, True))
comm = dpgo_amd.Comm(star.group, 0, 1)
star.initialize(X0)
for _ in range(3):
    star.step()
star.group.sync()
print("MARK begin", flush=True)
for _ in range(10):
    star.step()
star.group.sync()
print("MARK end", flush=True)
comm.close()
