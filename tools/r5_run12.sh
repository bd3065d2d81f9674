timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4
export DPGO_HOST_TIMING=1
for g in auto 0; do [ $g = auto ] && unset DPGO_ITER_GRAPH || export DPGO_ITER_GRAPH=$g
for i in 1 2; do timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 --windows 5 2>/tmp/e.err | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('emu $g %.4f ms' % j['ms_per_step'], j['graphs'])"; done; done
unset DPGO_HOST_TIMING DPGO_ITER_GRAPH
bash tools/final_profile.sh r05 rates 2>&1 | tail -3
python3 -c "
import json
for c in json.load(open('gpurun_out/final/r05_config_rates.json')): print(c['config'], round(c['gpu_iters_per_s']), [round(x) for x in c['gpu_iters_per_s_runs']])"
