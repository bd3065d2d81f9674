mkdir -p gpurun_out/r4
DPGO_AMD_LIB=$PWD/.ab/lib_nocontract.so python tools/probes/flow_diff.py > gpurun_out/r4/flow_diff.txt 2>&1
