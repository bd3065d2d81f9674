# what a starved host does in the long run: the emulated rank's timed window many times over, under tools/starve.py
tag=${1:-r5/st}; mkdir -p gpurun_out/$tag
export DPGO_HOST_TIMING=1
for rep in 1 2; do for w in spin auto; do for g in 0 1; do
  export DPGO_ITER_GRAPH=$g DPGO_WAIT=$w
  echo "== graph=$g wait=$w starve=7 rep=$rep"
  timeout 400 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 --windows 15 --starve-host 7 2>gpurun_out/$tag/emu.err | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); w=sorted(j['diagnostic_windows_ms_per_step']); s=j['diagnostic_starved_host']
print('emu mean %.4f ms  median window %.4f  best %.4f  worst %.4f' % (j['ms_per_step'], w[len(w)//2], w[0], w[-1]), j['graphs'])
print('   wall %.2f s, own cpu %.2f s, spinners cpu' % (s['wall_s'], s['own_cpu_s']), ['%.2f' % x for x in s['spinners_cpu_s']])"
  grep "^\[host\]" gpurun_out/$tag/emu.err
done; done; done 2>&1 | tee gpurun_out/$tag/summary.txt
