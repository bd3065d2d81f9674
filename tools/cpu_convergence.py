#!/usr/bin/env python3
"""The CPU side of the metric's second half: the C++ CPU restatement (tools/cpu_baseline/cpu_dpgo)
run on the headline instance until it has passed the reference objective -- iterations and seconds (iterate + update of all
nodes, communication excluded, as dist_pgo.cpp:496-521 times them) to come within 1e-6 (relative) of the lowest objective it
reaches -- on all the cores the box grants.  Output: one JSON object (committed once per round as profiles/rNN_cpu_convergence.json;
bench.py quotes it next to the GPU's `convergence` and measures the GPU's iterations / seconds to the SAME objective).

  python tools/cpu_convergence.py [iterations=260] > profiles/r03_cpu_convergence.json
"""
import json
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 260
    import bench
    import dpgo_amd
    from dpgo_amd import synthetic
    h = synthetic.HEADLINE
    g = synthetic.grid(50, 50, 40, 400000, seed=h["seed"])
    G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
    X0 = G.chordal_initialization()
    d, N, m = 3, g["num_poses"], len(g["I"])
    tmp = tempfile.mkdtemp(prefix="dpgo_cpu_")
    fe, fx = os.path.join(tmp, "edges.bin"), os.path.join(tmp, "X0.bin")
    rec = np.dtype([("i", "<i4"), ("j", "<i4"), ("R", "<f8", (d * d,)), ("t", "<f8", (d,)), ("kappa", "<f8"), ("tau", "<f8")])
    E = np.zeros(m, rec)
    E["i"], E["j"] = g["I"], g["J"]
    E["R"], E["t"] = np.asarray(g["R"]).reshape(m, d * d), g["t"]
    E["kappa"], E["tau"] = g["kappa"], g["tau"]
    with open(fe, "wb") as fh:
        fh.write(struct.pack("<iii", d, N, m))
        fh.write(E.tobytes())
    np.asfortranarray(X0, dtype=np.float64).T.copy().tofile(fx)
    cores = bench._host_cores()
    exe = os.path.join(ROOT, "tools", "cpu_baseline", "cpu_dpgo")
    t0 = time.time()
    out = subprocess.run([exe, fe, fx, "8", "1", str(iters), str(cores), "trace"], capture_output=True, text=True, check=True)
    wall = time.time() - t0
    for f in (fe, fx):
        os.remove(f)
    os.rmdir(tmp)
    rows = [l.replace(":", "").split() for l in out.stderr.splitlines() if l[:1].isdigit()]
    F = np.array([float(r[1]) for r in rows])
    T = np.array([float(r[2]) for r in rows])
    CG = np.array([int(r[3]) if len(r) > 3 else 0 for r in rows])
    best = float(F.min())
    hit = int(np.argmax(F <= best * (1 + 1e-6)))
    cpu_model = "unknown"
    try:
        cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    res = json.loads(out.stdout.strip().splitlines()[-1])
    print(json.dumps({
        "workload": "synthetic SE(3) lattice 50x50x40, 100000 poses / 400000 edges, huber loss, AMM-PGO#, num_nodes=8, chordal init",
        "tool": "tools/cpu_baseline/cpu_dpgo (C++ restatement, g++ -O3 -march=native -fopenmp), nodes dealt to the threads",
        "cpu_model": cpu_model, "cores": min(cores, 8), "host_cores_granted": cores,
        "iterations_run": iters, "lowest_2F": best, "iterations_to_1e-6": hit, "seconds_to_1e-6": float(T[hit]),
        "mean_s_per_iter_to_1e-6": float(T[hit]) / max(hit, 1), "seconds_whole_run": float(T[-1]), "wall_s_with_setup": wall,
        "setup_s": res["setup_s"], "cg_steps_to_1e-6": int(CG[hit]), "objective_2F_start": float(F[0]),
        "objective_2F_at": {str(k): float(F[k]) for k in (1, 10, 50, 100, 150, 200, 250) if k < len(F)},
        "timing_scope": "sum over iterations of iterate() + update() of all nodes; communication and set-up excluded (dist_pgo.cpp:496-521)",
    }))


if __name__ == "__main__":
    main()
