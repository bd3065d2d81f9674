# what a starved host does to the waits: emulated rank + city10000 under tools/starve.py, every wait mode
tag=${1:-r5/st}; mkdir -p gpurun_out/$tag
cat /proc/sys/kernel/sched_* 2>/dev/null | head -5; uname -r; nproc
for w in spin auto block; do for g in 1 0; do
  export DPGO_HOST_TIMING=1 DPGO_ITER_GRAPH=$g DPGO_WAIT=$w
  echo "== graph=$g wait=$w starve=7"
  timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 100 --warmup 10 --starve-host 7 2>gpurun_out/$tag/emu.err | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('emu %.4f ms' % j['ms_per_step'], j['graphs'])"
  grep "^\[host\]" gpurun_out/$tag/emu.err
  timeout 300 python tests/config_rates.py --no-oracle --starve-host 7 --only city10000 2>&1 >/dev/null | grep -E "config|host"
done; done 2>&1 | tee gpurun_out/$tag/summary.txt
