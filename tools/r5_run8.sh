tag=r5/run8; mkdir -p gpurun_out/$tag
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -25 > gpurun_out/$tag/tests_full.txt
tail -5 gpurun_out/$tag/tests_full.txt
DPGO_ITER_GRAPH=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_graphs.py tests/test_degenerate_graphs.py tests/test_gpu_tnt_ref.py tests/test_golden_traces.py tests/test_gpu_dchordal.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -15 > gpurun_out/$tag/tests_graph_forced.txt
tail -4 gpurun_out/$tag/tests_graph_forced.txt
export DPGO_HOST_TIMING=1
for rep in 1 2; do for g in auto 0 1; do
  [ $g = auto ] && unset DPGO_ITER_GRAPH || export DPGO_ITER_GRAPH=$g
  echo "== DPGO_ITER_GRAPH=$g"
  timeout 600 python tests/config_rates.py --no-oracle --repeat 3 2>&1 >/dev/null | grep -E "config|segments replayed" | sed 's/oracle.*//; s/.*segments replayed since the host was found to be the slower side:/      replayed:/'
  timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 2>gpurun_out/$tag/emu.err | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('emu %.4f ms' % j['ms_per_step'], j['graphs'])"
done; done 2>&1 | tee gpurun_out/$tag/policy.txt
