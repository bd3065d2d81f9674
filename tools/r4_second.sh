mkdir -p gpurun_out/r4
DPGO_SPD_TRACE=1 DPGO_AMD_LIB=$PWD/.ab/lib_trace.so python tools/spd_sweep.py DPGO_SPD_WIDE 96 --levels > gpurun_out/r4/levels_n1_trace.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_comm.py -m gpu -x -q > gpurun_out/r4/comm_tests.txt 2>&1
