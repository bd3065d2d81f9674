#!/bin/bash
# knob_sweep_conv.sh VAR v1 v2 ...: headline ms/step AND seconds to the CPU-reached objective, factor sizes and levels for each
# value of an environment knob (the merge depth trades the early regime against the interior one)
var=$1; shift
for v in "$@"; do
  env $var=$v timeout 900 python bench.py --no-cpu --no-prof --traffic off 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['solver']; c=j.get('convergence') or {}
print('$var=$v  %.4f ms/step  to objective %s s / %s it  nnz_tt %.1fM nnz_rr %.1fM levels %d/%d' % (j['ms_per_step'], c.get('seconds_to_1e-6'), c.get('iterations_to_1e-6'), s['nnz_tt']/1e6, s['nnz_rr']/1e6, s['levels_tt'], s['levels_rr']))"
done
