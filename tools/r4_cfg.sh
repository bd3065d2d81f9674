mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ds in city10000 torus3D; do
rm -rf gpurun_out/trace_cfg
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_cfg -- python3 tools/probes/config_one.py $ds 8 0 40 > gpurun_out/r4/cfg_$ds.log 2>&1
python3 tools/trace_busy.py gpurun_out/trace_cfg 4000 > gpurun_out/r4/cfg_${ds}_busy.txt
python3 tools/trace_tail.py gpurun_out/trace_cfg 400 > gpurun_out/r4/cfg_${ds}_timeline.txt
done
rm -rf gpurun_out/trace_cfg
python3 tools/probes/config_one.py city10000 8 0 40 > gpurun_out/r4/cfg_city_plain.log 2>&1
