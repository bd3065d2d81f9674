import os, sys, time
sys.path.insert(0, os.getcwd())
import dpgo_amd
for ds, nn in (("city10000", 8), ("torus3D", 8), ("sphere2500", 1)):
    path = os.path.join("fixtures", "g2o", ds + ".g2o")
    G = dpgo_amd.read_g2o(path, nn)
    X0 = G.chordal_initialization()
    gpu = dpgo_amd.DistPGO(G, dpgo_amd.Options.driver(0, True), X0=X0)
    for _ in range(5): gpu.step()
    dpgo_amd.prof_enable(True)
    n = 30
    t0 = time.perf_counter()
    for _ in range(n): gpu.step()
    gpu.group.sync()
    el = time.perf_counter() - t0
    st = dpgo_amd.prof_collect(); dpgo_amd.prof_enable(False)
    r = [gpu.group.results(k) for k in range(nn)]
    print(ds, "ms/iter (instrumented) %.2f" % (1e3 * el / n), "inner CG per iter (last)", sum(int(x.tnt_inner_iterations) for x in r), "refined", sum(int(x.refined) for x in r), gpu.group.solver_stats())
    tot = sum(v[0] for v in st.values())
    for k, (ms, by, cnt) in sorted(st.items(), key=lambda kv: -kv[1][0]):
        if cnt: print("   %-14s %7.3f ms/iter  %6.1f launches/iter  avg %5.1f us" % (k, ms / n, cnt / n, 1e3 * ms / cnt))
    print("   sum kernels %.3f ms/iter" % (tot / n))
