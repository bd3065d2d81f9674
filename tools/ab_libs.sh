# A/B of library builds on ONE box: bash tools/ab_libs.sh <tag> <reps> name1 name2 ...  (name = cur | a .ab/lib_<name>.so)
# every build runs the headline bench and the emulated 8-GPU rank <reps> times, interleaved; prints min and median
tag=$1; reps=$2; shift 2
mkdir -p gpurun_out/$tag
for rep in $(seq $reps); do for n in "$@"; do
  L=$PWD/.ab/lib_$n.so; [ $n = cur ] && L=$PWD/dpgo_amd/libdpgo_amd.so
  DPGO_AMD_LIB=$L timeout 300 python bench.py --no-cpu --no-prof --traffic off --converge 0 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys; print('n1 $n %.4f' % json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/$tag/raw.txt
  DPGO_AMD_LIB=$L timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 80 --warmup 10 2>/dev/null | python3 -c "
import json,sys; print('emu $n %.4f' % json.loads(sys.stdin.read())['ms_per_step'])" >> gpurun_out/$tag/raw.txt
done; done
python3 - <<PY
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/$tag/raw.txt"):
    k, n, v = l.split(); d[(k, n)].append(float(v))
for (k, n), v in sorted(d.items()):
    print("%-4s %-12s min %.4f median %.4f  (%s)" % (k, n, min(v), statistics.median(v), " ".join("%.4f" % x for x in v)))
PY
