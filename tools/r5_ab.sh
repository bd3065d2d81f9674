# same-box A/B of the host-side mechanisms on an idle host: segments eager / replayed / by measurement, waiting by polling /
# sleeping; the emulated rank of the 8-GPU run and the small parity configurations.  usage: bash tools/r5_ab.sh <tag> [reps]
tag=${1:-r5/ab}; reps=${2:-1}; mkdir -p gpurun_out/$tag
# (DPGO_WAIT=block, the sleeping wait, was removed in round 6: DESIGN 9)
for rep in $(seq $reps); do for g in 0 1 auto; do for w in spin; do
  [ $g = auto ] && [ $w = block ] && continue
  export DPGO_HOST_TIMING=1 DPGO_WAIT=$w
  [ $g = auto ] && unset DPGO_ITER_GRAPH || export DPGO_ITER_GRAPH=$g
  echo "== DPGO_ITER_GRAPH=$g DPGO_WAIT=$w rep=$rep"
  timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 --windows 5 2>gpurun_out/$tag/emu.err | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('emulated rank 3 of 8: %.4f ms / iteration' % j['ms_per_step'], j['graphs'])"
  grep "^\[host\] node" gpurun_out/$tag/emu.err
  timeout 600 python tests/config_rates.py --no-oracle --repeat 5 2>&1 >/dev/null | grep -E "config|segments replayed" | sed 's/oracle.*//; s/.*segments replayed since the host was found to be the slower side:/      replayed once the host was found to be the slower side:/'
done; done; done 2>&1 | tee gpurun_out/$tag/summary.txt
