mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/startr
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d gpurun_out/startr -- python3 tools/probes/star_api_trace.py > gpurun_out/startr.log 2>&1
python3 tools/probes/star_api_summary.py gpurun_out/startr > gpurun_out/final/r04_star_api_trace.txt 2>&1
rm -rf gpurun_out/startr
for v in 0 1; do DPGO_CG_GRAPH=$v python3 tools/probes/config_phase.py city10000 8 0 40; DPGO_CG_GRAPH=$v python3 tools/probes/config_phase.py sphere2500 1 0 100; done > gpurun_out/final/r04_cg_graph_ab.txt 2>&1
