# HIP API calls per iteration of the emulated rank (one node per GPU), eager against replayed segments
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=${1:-r5/api}; mkdir -p $R/gpurun_out/$tag
for g in 0 1; do for n in 15 30; do
  rm -rf /tmp/api_g${g}_$n
  DPGO_ITER_GRAPH=$g rocprofv3 --hip-runtime-trace --output-format csv -d /tmp/api_g${g}_$n -- python3 $R/bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps $n --warmup 10 > /dev/null 2>&1
done
echo "== DPGO_ITER_GRAPH=$g: emulated rank 3 of 8 (one node per GPU), iterations 25..40 of the run (early regime: one CG step per refinement)" > $R/gpurun_out/$tag/api_emulated_rank_graph$g.txt
python3 $R/tools/api_per_iteration.py /tmp/api_g${g}_15 15 /tmp/api_g${g}_30 30 >> $R/gpurun_out/$tag/api_emulated_rank_graph$g.txt
cat $R/gpurun_out/$tag/api_emulated_rank_graph$g.txt
done
