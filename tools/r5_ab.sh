# same-box A/B: iteration graphs off / on, host plain / starved; emulated rank + city10000 + sphere2500
tag=${1:-r5/ab}; mkdir -p gpurun_out/$tag
for rep in 1 2; do for g in 0 1; do for s in -1 7; do
  DPGO_HOST_TIMING=1 DPGO_ITER_GRAPH=$g timeout 600 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 --starve-host $s 2>gpurun_out/$tag/emu_g${g}_s${s}_$rep.err | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('emu graph=$g starve=$s %.4f ms' % j['ms_per_step'], j['graphs'])"
  grep "^\[host\]" gpurun_out/$tag/emu_g${g}_s${s}_$rep.err
  DPGO_HOST_TIMING=1 DPGO_ITER_GRAPH=$g timeout 600 python tests/config_rates.py --no-oracle --starve-host $s --only "city10000" 2>&1 >/dev/null | grep -E "config|host"
  DPGO_HOST_TIMING=1 DPGO_ITER_GRAPH=$g timeout 600 python tests/config_rates.py --no-oracle --starve-host $s --only "sphere2500" 2>&1 >/dev/null | grep -E "config|host"
done; done; done 2>&1 | tee gpurun_out/$tag/summary.txt
