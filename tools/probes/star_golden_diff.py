"""Device AMM-PGO* on M3500 / 4 nodes from the committed oracle warm start against the committed oracle trace: where do the
objectives part, and by how much (per 25 iterations: largest relative difference)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dpgo_amd
c = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_traces.json")))["cases"]["config5_M3500_star_distinit_4nodes"]
G = dpgo_amd.read_g2o(os.path.join(ROOT, "fixtures", "g2o", "M3500.g2o"), 4)
star = dpgo_amd.DPGOStar(G, dpgo_amd.Options.driver(0, True))
X0 = np.load(os.path.join(ROOT, "tests", "golden", "config5_M3500_star_distinit_4nodes_X0.npz"))["X0"]
assert star.initialize(X0) == 0
ref = np.asarray(c["trace_F"])
got, br = [star.state()["fobj"]], [0]
for _ in range(c["iterations"]):
    assert star.step() == 0
    st = star.state()
    got.append(st["fobj"]); br.append(st["branches"])
got = np.asarray(got)
rel = np.abs(got - ref) / np.abs(ref)
for k in range(0, len(ref), 25):
    print("iterations %3d..%3d  max rel diff %.2e   branches taken %s" % (k, min(k + 24, len(ref) - 1), rel[k:k + 25].max(), sorted(set(br[k:k + 25]))))
print("final: device %.12g oracle %.12g rel %.2e" % (got[-1], ref[-1], rel[-1]))
