#!/bin/bash
# what the neighbour-to-neighbour exchange costs an iteration at one node per GPU, short of the wire: the emulated rank with
# the p2p path run against itself, five windows per run, runs interleaved.  usage: bash tools/r6_xchg.sh <tag> cfg...
tag=$1; shift; mkdir -p gpurun_out/$tag
for rep in 1 2 3; do for cfg in "$@"; do for x in "" "--force-exchange"; do
  ( export $cfg; timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --traffic off --converge 0 --steps 60 --warmup 10 --windows 5 $x 2>>gpurun_out/$tag/err.txt | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); w=j.get('diagnostic_windows_ms_per_step') or [j['ms_per_step']]
print('%-28s %-18s min %.4f  windows %s  ready-to-done: %s' % ('$cfg', '$x' or '(no exchange)', min(w), ' '.join('%.4f' % v for v in w), j.get('exchange_us_ready_to_done')))" )
done; done; done 2>&1 | tee gpurun_out/$tag/self_exchange.txt
