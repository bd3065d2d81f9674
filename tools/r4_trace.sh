mkdir -p gpurun_out/r4
bash tools/trace_levels.sh r4_n1
bash tools/trace_levels.sh r4_emu --emulate-world 8 --emulate-rank 3
