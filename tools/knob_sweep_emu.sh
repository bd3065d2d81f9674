#!/bin/bash
# knob_sweep_emu.sh VAR v1 v2 ...: ms/step of one rank of an 8-GPU run (emulated on one GPU) for each value of an environment knob
var=$1; shift
for v in "$@"; do
  env $var=$v timeout 600 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['solver']
print('$var=$v  %.4f ms/step  nnz_tt %.1fM nnz_rr %.1fM levels %d/%d' % (j['ms_per_step'], s['nnz_tt']/1e6, s['nnz_rr']/1e6, s['levels_tt'], s['levels_rr']))"
done
