# same-box A/B of the host-side mechanisms: iteration graphs off / on, waiting by polling / adaptively, host plain / starved
# (tools/starve.py); the emulated rank of the 8-GPU run and two small parity configurations.  usage: bash tools/r5_ab.sh <tag> [reps]
tag=${1:-r5/ab}; reps=${2:-1}; mkdir -p gpurun_out/$tag
for rep in $(seq $reps); do for s in -1 7; do for g in 0 1; do for w in spin auto; do
  export DPGO_HOST_TIMING=1 DPGO_ITER_GRAPH=$g DPGO_WAIT=$w
  echo "== graph=$g wait=$w starve=$s rep=$rep"
  timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 --starve-host $s 2>gpurun_out/$tag/emu.err | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('emu %.4f ms' % j['ms_per_step'], j['graphs'])"
  grep "^\[host\]" gpurun_out/$tag/emu.err
  for c in city10000 sphere2500 M3500; do
    timeout 300 python tests/config_rates.py --no-oracle --starve-host $s --only $c 2>&1 >/dev/null | grep -E "config|host"
  done
done; done; done; done 2>&1 | tee gpurun_out/$tag/summary.txt
