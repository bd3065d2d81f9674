# kernel timelines of the emulated rank: eager against replayed segments (last dispatches of each run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=${1:-r5/prof}; mkdir -p $R/gpurun_out/$tag
for g in 0 1; do
  rm -rf /tmp/prof_g$g
  DPGO_ITER_GRAPH=$g rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g$g -- python3 $R/bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 30 --warmup 10 > /dev/null 2>&1
  python3 $R/tools/trace_tail.py /tmp/prof_g$g 150 > $R/gpurun_out/$tag/timeline_graph$g.txt
  python3 $R/tools/trace_busy.py /tmp/prof_g$g 2000 > $R/gpurun_out/$tag/busy_graph$g.txt
done
tail -3 $R/gpurun_out/$tag/busy_graph*.txt
