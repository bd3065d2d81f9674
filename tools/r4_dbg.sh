mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/probes/star_rate.py > gpurun_out/r4/star_busy.txt 2>&1
rm -rf gpurun_out/startr
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/startr -- python3 tools/probes/star_rate.py >> gpurun_out/r4/star_busy.txt 2>&1
python3 tools/trace_busy.py gpurun_out/startr 3000 >> gpurun_out/r4/star_busy.txt
python3 tools/trace_tail.py gpurun_out/startr 260 > gpurun_out/r4/star_timeline.txt
rm -rf gpurun_out/startr
