# usage: trace_levels.sh <tag> [bench args...]   -> gpurun_out/timeline_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$tag -- python3 bench.py --no-cpu --no-prof --steps 3 --warmup 2 "$@" > gpurun_out/trace_$tag.log 2>&1
python3 tools/trace_tail.py gpurun_out/trace_$tag 260 > gpurun_out/timeline_$tag.txt
rm -rf gpurun_out/trace_$tag
