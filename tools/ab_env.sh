# A/B of environment switches on ONE box (boxes differ by +-3 %): every setting twice, interleaved.
# Usage: bash tools/ab_env.sh <tag> "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...
tag=$1; args=$2; shift 2
for rep in 1 2; do
  i=0
  for setting in "$@"; do
    i=$((i+1))
    env $setting python bench.py $args --no-cpu --no-prof --converge 0 > gpurun_out/${tag}_${i}_${rep}.json 2> gpurun_out/${tag}_${i}_${rep}.err
    python - <<PY
import json
try:
    j = json.load(open("gpurun_out/${tag}_${i}_${rep}.json")); print("${setting} rep ${rep}: %.1f it/s %.4f ms" % (j["value"], j["ms_per_step"]))
except Exception as e: print("${setting} FAILED", e)
PY
  done
done
