#!/bin/bash
# bit-level A/B on the GPU box: the current library against a saved build (.ab/lib_<name>.so) and against itself with a
# switch set.  usage: bash tools/r6_bits.sh <tag> <base-lib-name> ["ENV=val ..." ...]
tag=$1; base=$2; shift 2
mkdir -p gpurun_out/$tag
DPGO_AMD_LIB=$PWD/.ab/lib_$base.so timeout 900 python tools/probes/ab_bits.py /tmp/bits_base.npz 2>gpurun_out/$tag/base.err || { echo "base run failed"; tail -5 gpurun_out/$tag/base.err; }
timeout 900 python tools/probes/ab_bits.py /tmp/bits_cur.npz 2>gpurun_out/$tag/cur.err || { echo "current run failed"; tail -20 gpurun_out/$tag/cur.err; }
echo "== current build against $base" | tee gpurun_out/$tag/bits.txt
python tools/probes/ab_bits.py --compare /tmp/bits_base.npz /tmp/bits_cur.npz 2>&1 | tee -a gpurun_out/$tag/bits.txt
i=0
for cfg in "$@"; do i=$((i+1))
  env $cfg timeout 900 python tools/probes/ab_bits.py /tmp/bits_$i.npz 2>gpurun_out/$tag/cfg$i.err || { echo "run with $cfg failed"; tail -20 gpurun_out/$tag/cfg$i.err; }
  echo "== current build with $cfg against $base" | tee -a gpurun_out/$tag/bits.txt
  python tools/probes/ab_bits.py --compare /tmp/bits_base.npz /tmp/bits_$i.npz 2>&1 | tee -a gpurun_out/$tag/bits.txt
done
