run() { echo "$*: $(env DYN_ONLY=1 "$@" timeout 200 python3 tools/probes/dynamic_headline.py 50,50,40,400000 40 2>&1 | grep -A1 'rescale=1' | cut -c1-200)"; }
python -m pytest tests/test_gpu_factor.py -m gpu -x -q 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "switches or dynamic or rescale" 2>&1 | tail -2
for rep in 1 2; do
run DPGO_X=1
run DPGO_SPD_EXTEND_SLOTS=1
done
