#!/bin/bash
# Same-box A/B on the GPU box (boxes of the pool differ by more than most effects): every configuration <reps> times,
# interleaved; prints min and median ms/step of the headline bench (n1), of the emulated 8-GPU rank (emu) and, with
# CONV=1, the seconds to the CPU-reached objective.  A configuration is a quoted list of NAME=value words; lib=<name>
# selects a library build .ab/lib_<name>.so of tools/build_variant.sh (lib=cur: the current one).
#   bash tools/ab.sh <tag> <reps> "lib=cur" "lib=base" ...                 A/B of library builds
#   bash tools/ab.sh <tag> <reps> "DPGO_X=0" "DPGO_X=1 DPGO_Y=2" ...       A/B of environment switches
#   for v in 1 2 3; do ...; done  (a knob sweep is an A/B with one configuration per value)
tag=$1; reps=$2; shift 2
mkdir -p gpurun_out/$tag; : > gpurun_out/$tag/raw.txt
for rep in $(seq $reps); do i=0; for cfg in "$@"; do i=$((i+1))
  envs=(); lib=$PWD/dpgo_amd/libdpgo_amd.so
  for w in $cfg; do case $w in lib=cur) ;; lib=*) lib=$PWD/.ab/lib_${w#lib=}.so ;; *) envs+=("$w") ;; esac; done
  run() { env DPGO_AMD_LIB=$lib "${envs[@]}" timeout 600 python bench.py --no-cpu --no-prof --traffic off "$@" 2>/dev/null; }
  if [ "$CONV" = 1 ]; then
    run --steps 20 --warmup 5 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['convergence']; print('n1 $i %.4f' % j['ms_per_step']); print('conv $i %.4f' % c['seconds_to_1e-6'])" >> gpurun_out/$tag/raw.txt
  else
    run --converge 0 --steps 40 --warmup 10 | python3 -c "
import json,sys; print('n1 $i %.4f' % json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/$tag/raw.txt
  fi
  run --converge 0 --emulate-world 8 --emulate-rank 3 --steps 60 --warmup 10 | python3 -c "
import json,sys; print('emu $i %.4f' % json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/$tag/raw.txt
done; done
python3 - "$@" <<PY
import collections, statistics, sys
cfgs = sys.argv[1:]
d = collections.defaultdict(list)
for l in open("gpurun_out/$tag/raw.txt"):
    k, n, v = l.split(); d[(k, int(n))].append(float(v))
for (k, n), v in sorted(d.items()):
    print("%-4s %-40s min %.4f median %.4f  (%s)" % (k, cfgs[n - 1][:40], min(v), statistics.median(v), " ".join("%.4f" % x for x in v)))
PY
