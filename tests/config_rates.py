#!/usr/bin/env python3
"""Outer iterations per second of the BASELINE.json parity configurations: the HIP path on one GPU and the oracle on
one host core of the same box (same initial point, same number of iterations).  Small graphs are launch-latency
bound on a GPU; the table says by how much.  Usage (GPU box): python tests/config_rates.py > gpurun_out/config_rates.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse                                                  # noqa: E402
_ap = argparse.ArgumentParser()
_ap.add_argument("--starve-host", type=int, default=-1, help="pin to one core with this many spinning siblings (tools/starve.py)")
_ap.add_argument("--only", type=str, default="", help="substring of the configuration names to run")
_ap.add_argument("--no-oracle", action="store_true", help="skip the oracle's side of the table")
_ap.add_argument("--repeat", type=int, default=1, help="run this many times the iterations of the table (GPU side)")
_args = _ap.parse_args()
if _args.starve_host >= 0:      # (before anything touches the GPU)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import starve
    starve.prepare(_args.starve_host)     # (the spinners start with the first timed loop)
import dpgo_amd                                                  # noqa: E402
from oracle import g2o as og                                     # noqa: E402
from oracle.hash import Options as OOptions                      # noqa: E402
from oracle.star import DistPGO as ODistPGO, DPGOStar as ODPGOStar, chordal_initialization   # noqa: E402

CASES = [  # name, dataset, nodes, loss, accelerated, scheme, iterations
    ("config 1: smallGrid3D, MM-PGO, 2 nodes", "smallGrid3D", 2, 0, False, "hash", 200),
    ("config 2: sphere2500, AMM-PGO#, 1 node", "sphere2500", 1, 0, True, "hash", 200),
    ("config 3: torus3D, AMM-PGO#, 8 nodes", "torus3D", 8, 0, True, "hash", 60),
    ("config 3: city10000 (SE2), AMM-PGO#, 8 nodes", "city10000", 8, 0, True, "hash", 40),
    ("config 5: M3500 (SE2), AMM-PGO*, 4 nodes", "M3500", 4, 0, True, "star", 40),
]
out = []
for name, ds, nn, loss, acc, scheme, iters in CASES:
    if _args.only and _args.only not in name:
        continue
    path = os.path.join(ROOT, "fixtures", "g2o", ds + ".g2o")
    num_poses, mm = og.read_g2o_file(path)
    X0 = chordal_initialization(num_poses, mm)
    G = dpgo_amd.read_g2o(path, nn)
    if scheme == "hash":
        gpu = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(loss, acc), X0=X0)
        orc = ODistPGO(path, nn, OOptions.driver(loss, acc), X0=X0, mm=mm, num_poses=num_poses)
        gstep, ostep = gpu.step, (lambda: orc.step(evaluate=False))
    else:
        gpu = dpgo_amd.DPGOStar(G, dpgo_amd.Options.driver(loss, acc))
        gpu.initialize(X0)
        orc = ODPGOStar(path, nn, OOptions.driver(loss, acc), mm=mm, num_poses=num_poses)
        orc.initialize(X0)
        gstep, ostep = gpu.step, orc.step
    if _args.starve_host >= 0:
        starve.release()
        time.sleep(0.2)
    for _ in range(3):
        gstep()
    gpu.group.sync()
    t0 = time.perf_counter()
    for _ in range(iters * _args.repeat):
        gstep()
    gpu.group.sync()
    tg = (time.perf_counter() - t0) / _args.repeat
    t0 = time.perf_counter()
    for _ in range(0 if _args.no_oracle else iters):
        ostep()
    to = max(time.perf_counter() - t0, 1e-9)
    out.append({"config": name, "poses": num_poses, "edges": len(mm), "iterations": iters, "gpu_iterations_timed": iters * _args.repeat,
                "gpu_iters_per_s": iters / tg, "oracle_1core_iters_per_s": iters / to})
    print("%-48s GPU %8.1f it/s   oracle %7.2f it/s" % (name, iters / tg, iters / to), file=sys.stderr)
print(json.dumps(out))
