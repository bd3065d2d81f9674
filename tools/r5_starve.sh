# what a starved host does in the long run: the emulated rank's timed window 15 times over with the process pinned to ONE core
# shared with 7 busy-looping siblings (tools/starve.py; the line gives the CPU seconds this process and every sibling got:
# a run whose siblings got a few hundredths of a second each has escaped the pin -- the pool's boxes rewrite cpusets now and
# then -- and is not a starved run)
tag=${1:-r5/st}; mkdir -p gpurun_out/$tag
export DPGO_HOST_TIMING=1
# (DPGO_WAIT=block, the sleeping wait, was removed in round 6: DESIGN 9)
for rep in 1 2; do for w in spin; do for g in 0 1; do
  export DPGO_ITER_GRAPH=$g DPGO_WAIT=$w
  echo "== DPGO_ITER_GRAPH=$g DPGO_WAIT=$w, 7 spinning siblings on the one core, rep=$rep"
  timeout 400 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 --windows 15 --starve-host 7 2>gpurun_out/$tag/emu.err | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); w=sorted(j['diagnostic_windows_ms_per_step']); s=j['diagnostic_starved_host']
print('emulated rank 3 of 8: mean %.4f ms / iteration  median window %.4f  best %.4f  worst %.4f' % (j['ms_per_step'], w[len(w)//2], w[0], w[-1]), j['graphs'])
print('   wall %.2f s, own cpu %.2f s, siblings cpu' % (s['wall_s'], s['own_cpu_s']), ['%.2f' % x for x in s['spinners_cpu_s']])"
  grep "^\[host\]" gpurun_out/$tag/emu.err
done; done; done 2>&1 | tee gpurun_out/$tag/summary.txt
