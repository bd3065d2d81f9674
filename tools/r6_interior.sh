#!/bin/bash
# where the time of the interior regime goes (the headline run to its objective: 1-3 nodes still in the CG, 20+ steps each):
# kernel trace of a run of N iterations, the window [from, to) of its dispatches by kernel name
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=${1:-r6/interior}; mkdir -p $R/gpurun_out/$tag
rm -rf /tmp/prof_int
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_int -- python3 $R/bench.py ${BENCH_ARGS} --no-cpu --no-prof --traffic off --converge 0 --steps ${STEPS:-110} --warmup 10 > /dev/null 2>&1
python3 $R/tools/trace_window.py /tmp/prof_int ${FROM:-0.6} ${TO:-0.95} | tee $R/gpurun_out/$tag/window.txt
python3 $R/tools/trace_grids.py /tmp/prof_int "k_spd_level<3, 3" ${FROM:-0.6} ${TO:-0.95} | tee $R/gpurun_out/$tag/grids.txt
