#!/bin/bash
# knob_sweep2.sh VAR v1 v2 ...: early-regime ms/step AND seconds to the reference objective for each value of an environment knob
var=$1; shift
for v in "$@"; do
  env $var=$v timeout 900 python bench.py --no-cpu --no-prof --converge 250 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['solver']; c=j['convergence']
print('$var=$v  %.3f ms/step  %d iterations %.3f s to 1e-6  nnz %.1fM / %.1fM levels %d/%d' % (j['ms_per_step'], c['iterations_to_1e-6'], c['seconds_to_1e-6'], s['nnz_tt']/1e6, s['nnz_rr']/1e6, s['levels_tt'], s['levels_rr']))"
done
