mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_comm.py tests/test_star_multiprocess.py tests/test_gpu_dchordal.py tests/test_dchordal_multiprocess.py tests/test_fuzz_graphs.py -m gpu -x -q > gpurun_out/r4/star_tests_full.txt 2>&1
grep -E "passed|failed" gpurun_out/r4/star_tests_full.txt > gpurun_out/r4/star_tests.txt
