# On the GPU box: the parity tests that exercise the iteration (fast subset), then the three headline numbers.
# Usage: bash tools/quick_check.sh <tag> [full]
tag=${1:-x}
mkdir -p gpurun_out/$(dirname $tag)
if [ "$2" = "full" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${tag}_tests.txt
else
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_graphs.py tests/test_degenerate_graphs.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${tag}_tests.txt
fi
cat gpurun_out/${tag}_tests.txt
timeout 300 python bench.py --no-cpu > gpurun_out/${tag}_n1.json 2> gpurun_out/${tag}_n1.err
timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu > gpurun_out/${tag}_emu.json 2> gpurun_out/${tag}_emu.err
python - <<PY
import json
for f in ("gpurun_out/${tag}_n1.json", "gpurun_out/${tag}_emu.json"):
    try:
        j = json.load(open(f))
        c = j.get("convergence") or {}
        print(f, "it/s %.1f ms %.4f frac %.3f conv_s %s conv_it %s" % (j["value"], j["ms_per_step"], (j["roofline"] or {}).get("frac", 0), c.get("seconds_to_1e-6"), c.get("iterations_to_1e-6")))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 gpurun_out/${tag}_n1.err
