# round 4, first GPU call: baseline on this round's box + the probes behind the plan
mkdir -p gpurun_out/r4
./tools/probes/boundary_probe > gpurun_out/r4/boundary_probe.txt 2>&1
python tools/spd_sweep.py DPGO_SPD_WIDE 96 --one --levels > gpurun_out/r4/levels_one_default.txt 2>&1
DPGO_AMD_LIB=$PWD/.ab/lib_dense.so python tools/spd_sweep.py DPGO_SPD_WIDE 96 --one --levels > gpurun_out/r4/levels_one_dense.txt 2>&1
DPGO_SPD_TRACE=1 DPGO_AMD_LIB=$PWD/.ab/lib_trace.so python tools/spd_sweep.py DPGO_SPD_WIDE 96 --one --levels > gpurun_out/r4/levels_one_trace.txt 2>&1
python tools/spd_sweep.py DPGO_SPD_FINE_ROOT8 0 --one --levels > gpurun_out/r4/levels_one_root16.txt 2>&1
DPGO_AMD_LIB=$PWD/.ab/lib_dense.so python tools/spd_sweep.py DPGO_SPD_WIDE 96 --levels > gpurun_out/r4/levels_n1_dense.txt 2>&1
bash tools/quick_check.sh r4/base
