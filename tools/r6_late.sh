#!/bin/bash
# The parity configurations, four times as long as tools/r6_bits.sh runs them, with the host 0 / 40 / 250 us late to every
# read-back, eager and replayed: every run must leave the same bits.  usage: bash tools/r6_late.sh <tag>
tag=${1:-r6/late}; mkdir -p gpurun_out/$tag; export AB_ITERS_SCALE=${AB_ITERS_SCALE:-4}
run() { n=$1; shift; env "$@" timeout 1500 python tools/probes/ab_bits.py /tmp/late_$n.npz 2>gpurun_out/$tag/run$n.err || { echo "run $n ($*) failed"; tail -5 gpurun_out/$tag/run$n.err; }; }
run 0 A=1
i=0
for cfg in "DPGO_DEBUG_LATE_HOST_US=40" "DPGO_DEBUG_LATE_HOST_US=250" "DPGO_ITER_GRAPH=0" "DPGO_ITER_GRAPH=0 DPGO_DEBUG_LATE_HOST_US=120" "DPGO_ITER_GRAPH=1" "DPGO_ITER_GRAPH=1 DPGO_DEBUG_LATE_HOST_US=120" "DPGO_CG_GRAPH=0 DPGO_DEBUG_LATE_HOST_US=60"; do i=$((i+1))
  run $i $cfg
  echo "== $cfg against the defaults" | tee -a gpurun_out/$tag/late.txt
  python tools/probes/ab_bits.py --compare /tmp/late_0.npz /tmp/late_$i.npz 2>&1 | tee -a gpurun_out/$tag/late.txt
done
