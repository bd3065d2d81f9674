#!/bin/bash
# kernel timelines of the emulated rank (one node per GPU), one per configuration ("ENV=val ..." words; "" = defaults):
# the last dispatches of a short run + the summed kernel time per name.  usage: bash tools/r6_prof.sh <tag> <n-last> cfg...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; n=$2; shift 2; mkdir -p $R/gpurun_out/$tag
i=0
for cfg in "$@"; do i=$((i+1))
  rm -rf /tmp/prof_$i
  ( export $cfg DPGO_ITER_GRAPH=0; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$i -- python3 $R/bench.py ${BENCH_ARGS:---emulate-world 8 --emulate-rank 3} --no-cpu --no-prof --traffic off --converge 0 --steps 30 --warmup 10 > /dev/null 2>&1 )
  { echo "== $cfg"; python3 $R/tools/trace_tail.py /tmp/prof_$i $n; } > $R/gpurun_out/$tag/timeline_$i.txt
  { echo "== $cfg"; python3 $R/tools/trace_iter.py /tmp/prof_$i 20; } > $R/gpurun_out/$tag/iter_$i.txt
done
