# per-launch solve profile (DPGO_SPD_DUMP) for N=1 and the emulated rank -> gpurun_out/spd_<tag>_{n1,emu8}.txt
tag=$1
export DPGO_SPD_DUMP=1
python bench.py --no-cpu --no-prof --steps 2 --warmup 1 2>&1 | grep "^\[spd\]" > gpurun_out/spd_${tag}_n1.txt
python bench.py --no-cpu --no-prof --steps 2 --warmup 1 --emulate-world 8 --emulate-rank 3 2>&1 | grep "^\[spd\]" > gpurun_out/spd_${tag}_emu8.txt
