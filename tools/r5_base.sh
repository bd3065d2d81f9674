# round-5 baseline on one box: parity subset, headline, emulated rank (plain and with a starved host), parity-config rates
tag=${1:-r5/base}
mkdir -p gpurun_out/$tag
bash tools/quick_check.sh $tag/qc
for s in -1 7; do
  timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 200 --warmup 20 --starve-host $s > gpurun_out/$tag/emu_starve$s.json 2>/dev/null
  timeout 600 python tests/config_rates.py --no-oracle --starve-host $s > gpurun_out/$tag/rates_starve$s.json 2> gpurun_out/$tag/rates_starve$s.txt
done
python3 - <<PY
import json
for s in (-1, 7):
    try:
        j = json.load(open("gpurun_out/$tag/emu_starve%d.json" % s)); print("emu starve", s, "%.4f ms" % j["ms_per_step"])
    except Exception as e: print("emu starve", s, "FAILED", e)
    print(open("gpurun_out/$tag/rates_starve%d.txt" % s).read())
PY
