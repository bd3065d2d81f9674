"""tools/cpu_baseline/cpu_dpgo (the C++ CPU restatement bench.py times as `cpu_baseline`) against the oracle: same
per-iteration objective trace on smallGrid3D / Huber / AMM-PGO# / 2 nodes.  (The tool is a measurement aid, not part of
the product; this test keeps its numbers meaningful.)"""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import g2o as og
from oracle.hash import Options as OOptions
from oracle.star import DistPGO as ODistPGO, chordal_initialization

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "cpu_baseline", "cpu_dpgo")


@pytest.mark.parametrize("name,nn,loss,iters", [("smallGrid3D", 2, 1, 25), ("tinyGrid3D", 2, 3, 15)])
def test_cpu_tool_follows_the_oracle(fixtures_dir, tmp_path, name, nn, loss, iters):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    path = os.path.join(fixtures_dir, name + ".g2o")
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    d = mm.d
    fe, fx = str(tmp_path / "edges.bin"), str(tmp_path / "X0.bin")
    with open(fe, "wb") as fh:
        fh.write(struct.pack("<iii", d, num_poses, len(mm)))
        for e in range(len(mm)):
            fh.write(struct.pack("<ii", int(mm.ipose[e]), int(mm.jpose[e])))
            fh.write(np.concatenate([mm.R[e].ravel(), mm.t[e], [mm.kappa[e], mm.tau[e]]]).astype("<f8").tobytes())
    np.asfortranarray(X0).T.copy().tofile(fx)
    out = subprocess.run([EXE, fe, fx, str(nn), str(loss), str(iters), "2", "trace"], capture_output=True, text=True,
                         check=True, timeout=300).stderr
    got = np.array([float(l.split()[1]) for l in out.splitlines() if l[:1].isdigit()])
    orc = ODistPGO(path, nn, OOptions.driver(loss, True), X0=X0, mm=mm, num_poses=num_poses)
    # the tool prints 2 sum_a fobj^a (the per-node surrogate values, which carry kappa I for the intra edges where the
    # global evaluate_f has kappa R R^T: the fixture's quaternions are not normalised to machine precision)
    ref = [2 * sum(nd.results.fobj[0] for nd in orc.nodes)]
    for _ in range(iters):
        orc.step(evaluate=False)
        ref.append(2 * sum(nd.results.fobj[0] for nd in orc.nodes))
    ref = np.array(ref)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-6)
