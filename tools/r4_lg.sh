mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_degenerate_graphs.py tests/test_fuzz_graphs.py tests/test_gpu_tnt_ref.py tests/test_gpu_dchordal.py tests/test_cpu_baseline_tool.py -m gpu -x -q > gpurun_out/r4/lg_tests_full.txt 2>&1
grep -E "passed|failed|Error" gpurun_out/r4/lg_tests_full.txt | tail -5 > gpurun_out/r4/lg_tests.txt
rm -f gpurun_out/r4/lg_ab.txt
for rep in 1 2; do for v in 0 1; do
DPGO_SPD_ROOT_FINE_LIVE=$v timeout 300 python bench.py --no-cpu --no-prof --traffic off --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['convergence']; print('root_fine=$v n1 %.4f ms/step conv %.3f s %d it whole-run %.2f ms/it' % (j['ms_per_step'], c['seconds_to_1e-6'], c['iterations_to_1e-6'], c['mean_ms_per_iter_whole_run']))" >> gpurun_out/r4/lg_ab.txt
done; done
