#!/bin/bash
# lib_ab.sh NAME...: per-factor solve totals (tools/spd_sweep.py) for the A/B libraries .ab/lib_NAME.so of tools/build_variant.sh
for n in "$@"; do
  echo "#### $n"
  DPGO_AMD_LIB=$PWD/.ab/lib_$n.so python tools/spd_sweep.py DPGO_SPD_WIDE 96 $LIB_AB_FLAGS 2>&1 | grep -v "^=="
done
