mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_factor.py -m gpu -x -q -k one_triangle 2>&1 | tail -3 > gpurun_out/r4/sym_tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "engineering_switches or headline_size" 2>&1 | tail -3 >> gpurun_out/r4/sym_tests.txt
python tools/spd_sweep.py DPGO_SPD_ROOT_SYM 0 1 --levels > gpurun_out/r4/sym_levels_n1.txt 2>&1
python tools/spd_sweep.py DPGO_SPD_ROOT_SYM 0 1 --one --levels > gpurun_out/r4/sym_levels_one.txt 2>&1
for rep in 1 2; do for v in 0 1; do
DPGO_SPD_ROOT_SYM=$v timeout 300 python bench.py --no-cpu --no-prof --traffic off --converge 0 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('n1 sym=$v rep=$rep %.4f ms/step %.1f it/s' % (j['ms_per_step'], j['value']))" >> gpurun_out/r4/sym_ab.txt
DPGO_SPD_ROOT_SYM=$v timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('emu sym=$v rep=$rep %.4f ms/step' % (j['ms_per_step']))" >> gpurun_out/r4/sym_ab.txt
done; done
