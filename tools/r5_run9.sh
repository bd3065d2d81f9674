bash tools/r5_xchg.sh r5/xchg1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "strict_subset or engineering_switches" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8
DPGO_SETUP_TIMING=1 timeout 300 python tools/probes/dynamic_headline.py 50,50,40,400000 12 2>&1 | grep -E "rescale: [0-9]|iterations" | head -30
