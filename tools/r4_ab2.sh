mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_factor.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r4/ab2_tests.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_degenerate_graphs.py -m gpu -x -q 2>&1 | tail -3 >> gpurun_out/r4/ab2_tests.txt
rm -f gpurun_out/r4/ab2.txt
for rep in 1 2; do for lib in prev cur; do
L=$PWD/dpgo_amd/libdpgo_amd.so; [ $lib = prev ] && L=$PWD/.ab/lib_prev.so
DPGO_AMD_LIB=$L timeout 300 python bench.py --no-cpu --no-prof --traffic off --converge 0 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('n1 $lib rep=$rep %.4f ms/step %.1f it/s' % (j['ms_per_step'], j['value']))" >> gpurun_out/r4/ab2.txt
DPGO_AMD_LIB=$L timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('emu $lib rep=$rep %.4f ms/step' % (j['ms_per_step']))" >> gpurun_out/r4/ab2.txt
done; done
DPGO_SPD_FINE_ROOT=256 timeout 300 python bench.py --no-cpu --no-prof --traffic off --converge 0 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('n1 cur FINE_ROOT=256 %.4f ms/step %.1f it/s' % (j['ms_per_step'], j['value']))" >> gpurun_out/r4/ab2.txt
for lib in prev cur; do
L=$PWD/dpgo_amd/libdpgo_amd.so; [ $lib = prev ] && L=$PWD/.ab/lib_prev.so
DPGO_AMD_LIB=$L timeout 300 python bench.py --no-cpu --no-prof --traffic off --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['convergence']; print('conv $lib %.3f s %d it' % (c['seconds_to_1e-6'], c['iterations_to_1e-6']))" >> gpurun_out/r4/ab2.txt
done
