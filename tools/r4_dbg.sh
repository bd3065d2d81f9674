mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k engineering_switches 2>&1 | grep -B30 "AssertionError" | head -80 > gpurun_out/r4/dbg.txt
