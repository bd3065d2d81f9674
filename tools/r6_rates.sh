#!/bin/bash
# the parity configurations' rates (tests/config_rates.py --repeat 5), the current build against a saved one, interleaved
# usage: bash tools/r6_rates.sh <tag> <base-lib-name> [reps]
tag=$1; base=$2; reps=${3:-3}; mkdir -p gpurun_out/$tag; : > gpurun_out/$tag/rates.txt
for rep in $(seq $reps); do for lib in cur $base; do
  l=$PWD/dpgo_amd/libdpgo_amd.so; [ $lib != cur ] && l=$PWD/.ab/lib_$lib.so
  DPGO_AMD_LIB=$l timeout 600 python tests/config_rates.py --no-oracle --repeat ${REPEAT:-5} 2>&1 >/dev/null | grep "^config" | sed "s/oracle.*//; s/^/$lib  /" >> gpurun_out/$tag/rates.txt
done; done
sort -k2,3 -s gpurun_out/$tag/rates.txt | sort -t: -k1,1 -s | awk '{print}' > /dev/null
python3 - <<PY
import collections, statistics
d = collections.defaultdict(list)
for l in open("gpurun_out/$tag/rates.txt"):
    lib, rest = l.split(None, 1)
    name = rest.split("GPU")[0].strip(); v = float(rest.split("GPU")[1].split()[0])
    d[(name, lib)].append(v)
for (name, lib), v in sorted(d.items()):
    print("%-48s %-4s median %8.1f it/s  (%s)" % (name, lib, statistics.median(v), " ".join("%.0f" % x for x in v)))
PY
