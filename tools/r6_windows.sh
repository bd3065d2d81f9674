#!/bin/bash
# early-regime A/B of single switches: the same window (iterations 10..70 from the chordal start) five times per run, runs
# interleaved.  usage: bash tools/r6_begin.sh <bench args...> -- cfg...
args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
for rep in 1 2 3; do for cfg in "$@"; do
  ( export $cfg; timeout 300 python bench.py "${args[@]}" --no-cpu --no-prof --traffic off --converge 0 --steps 60 --warmup 10 --windows 5 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); w=j.get('diagnostic_windows_ms_per_step') or [j['ms_per_step']]; print('%-50s min %.4f  all %s' % ('$cfg', min(w), ' '.join('%.4f' % x for x in w)))" )
done; done
