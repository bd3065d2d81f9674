# round-5 check on one box: parity subset (optional), emulated rank and parity-config rates, plain and with a starved host
# usage: bash tools/r5_check.sh <tag> [tests|notests] [spinners]
tag=${1:-r5/x}; spin=${3:-7}
mkdir -p gpurun_out/$tag
if [ "$2" = "tests" ]; then
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_graphs.py tests/test_degenerate_graphs.py tests/test_gpu_tnt_ref.py tests/test_gpu_comm.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/$tag/tests.txt
  cat gpurun_out/$tag/tests.txt
fi
for s in -1 $spin; do
  timeout 600 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 --starve-host $s > gpurun_out/$tag/emu_starve$s.json 2>gpurun_out/$tag/emu_starve$s.err
  timeout 900 python tests/config_rates.py --no-oracle --starve-host $s > gpurun_out/$tag/rates_starve$s.json 2> gpurun_out/$tag/rates_starve$s.txt
done
timeout 300 python bench.py --no-cpu --no-prof --traffic off --converge 0 --steps 40 --warmup 10 > gpurun_out/$tag/n1.json 2>/dev/null
python3 - <<PY
import json
for s in (-1, $spin):
    try:
        j = json.load(open("gpurun_out/$tag/emu_starve%d.json" % s)); print("emu starve", s, "%.4f ms" % j["ms_per_step"], j.get("graphs"))
    except Exception as e: print("emu starve", s, "FAILED", e)
    print(open("gpurun_out/$tag/rates_starve%d.txt" % s).read())
try:
    j = json.load(open("gpurun_out/$tag/n1.json")); print("n1 %.4f ms" % j["ms_per_step"])
except Exception as e: print("n1 FAILED", e)
PY
