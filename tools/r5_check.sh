# round-5 check on one box: parity subset (optional), emulated rank and parity-config rates, plain and with a starved host
# usage: bash tools/r5_check.sh <tag> [tests|notests] [spinners]
tag=${1:-r5/x}; spin=${3:-7}
mkdir -p gpurun_out/$tag
if [ "$2" = "tests" ]; then
  timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_graphs.py tests/test_degenerate_graphs.py tests/test_gpu_tnt_ref.py tests/test_gpu_comm.py tests/test_golden_traces.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -15 > gpurun_out/$tag/tests.txt
  cat gpurun_out/$tag/tests.txt
fi
export DPGO_HOST_TIMING=1
for s in -1 $spin; do
  for g in 0 1; do
  DPGO_ITER_GRAPH=$g timeout 600 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 --starve-host $s 2>gpurun_out/$tag/emu.err | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('emu graph=$g starve=$s %.4f ms' % j['ms_per_step'], j['graphs'])"
  grep "^\[host\]" gpurun_out/$tag/emu.err
  done
  timeout 900 python tests/config_rates.py --no-oracle --starve-host $s 2>&1 > gpurun_out/$tag/rates_starve$s.json | grep -E "config|host"
done
timeout 300 python bench.py --no-cpu --no-prof --traffic off --converge 0 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('n1 %.4f ms' % j['ms_per_step'])"
