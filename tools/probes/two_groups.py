#!/usr/bin/env python3
"""Experiment: do two node groups on two streams (driven by two host threads) overlap each other's latency-bound
launches?  Headline graph, 8 nodes on one GPU: one group of 8 vs two groups of 4 (neighbours across groups frozen, as
in bench.py --emulate-world: timing only)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dpgo_amd
from dpgo_amd import synthetic

g = synthetic.grid(50, 50, 40, 400000, seed=synthetic.HEADLINE["seed"])
G = dpgo_amd.graph_from_edges(3, g["num_poses"], g["I"], g["J"], g["R"], g["t"], g["kappa"], g["tau"], 8)
opt = dpgo_amd.Options.driver(1, True)
X0 = G.chordal_initialization()
splits = [[list(range(8))], [[0, 1, 2, 3], [4, 5, 6, 7]], [[0, 1], [2, 3], [4, 5], [6, 7]]]
if len(sys.argv) > 1:
    splits = [splits[int(a)] for a in sys.argv[1:]]
for split in splits:
    grps = [dpgo_amd.NodeGroup(G, nodes, opt) for nodes in split]
    for q in grps:
        q.initialize_global(X0)
        q.update()

    def run(q, steps):
        for _ in range(steps):
            q.iterate()
            q.communicate_local()
            q.update()

    for q in grps:
        run(q, 5)
    for q in grps:
        q.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        for q in grps:
            q.iterate()
        for q in grps:
            q.communicate_local()
            q.update()
    for q in grps:
        q.sync()
    t_seq = (time.perf_counter() - t0) / 20
    th = [threading.Thread(target=run, args=(q, 20)) for q in grps]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    for q in grps:
        q.sync()
    t_par = (time.perf_counter() - t0) / 20
    print("groups %s: one host thread %.3f ms / iteration, one thread per group %.3f ms / iteration" % (split, t_seq * 1e3, t_par * 1e3))
    sys.stdout.flush()
    del grps
