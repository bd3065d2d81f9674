"""base run against DPGO_SPD_FLOW=1 on sphere2500 / 4 nodes: the largest difference of the iterates (should be 0)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = """
import sys, numpy as np
sys.path.insert(0, %r)
import dpgo_amd
G = dpgo_amd.read_g2o(%r, 4)
drv = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(1, True))
for it in range(int(sys.argv[2])):
    assert drv.step() == 0
np.save(sys.argv[1], drv.X())
""" % (ROOT, os.path.join(ROOT, "fixtures", "g2o", "sphere2500.g2o"))
tmp = tempfile.mkdtemp()
def run(tag, iters, **env):
    p = os.path.join(tmp, tag + ".npy")
    subprocess.check_call([sys.executable, "-c", code, p, str(iters)], env=dict(os.environ, **env))
    return np.load(p)
for iters in (1, 5, 25):
    a, b = run("a", iters), run("b", iters, DPGO_SPD_FLOW="1")
    print("iters", iters, "max diff", np.abs(a - b).max(), "lib", os.environ.get("DPGO_AMD_LIB", "default"))
