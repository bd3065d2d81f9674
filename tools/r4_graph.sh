mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_degenerate_graphs.py tests/test_fuzz_graphs.py tests/test_gpu_tnt_ref.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4/graph_tests.txt
rm -f gpurun_out/r4/graph_ab.txt
for rep in 1 2 3; do for v in 0 1; do
DPGO_CG_GRAPH=$v python3 tools/probes/config_one.py city10000 8 0 40 2>&1 | grep "it/s" | cut -c1-90 >> gpurun_out/r4/graph_ab.txt
DPGO_CG_GRAPH=$v python3 tools/probes/config_one.py sphere2500 1 0 100 2>&1 | grep "it/s" | cut -c1-90 >> gpurun_out/r4/graph_ab.txt
DPGO_CG_GRAPH=$v python3 tools/probes/config_one.py torus3D 8 1 60 2>&1 | grep "it/s" | cut -c1-90 >> gpurun_out/r4/graph_ab.txt
done; done
