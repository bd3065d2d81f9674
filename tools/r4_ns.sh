mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/ns_ab.txt
for rep in 1 2; do for v in "0 0" "0 1" "1 1"; do set -- $v
DPGO_CG_NODE_STREAMS=$1 DPGO_CG_GRAPH=$2 timeout 300 python bench.py --no-cpu --no-prof --traffic off --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['convergence']; print('node_streams=$1 graph=$2 n1 %.4f ms/step conv %.3f s %d it whole-run %.2f ms/it' % (j['ms_per_step'], c['seconds_to_1e-6'], c['iterations_to_1e-6'], c['mean_ms_per_iter_whole_run']))" >> gpurun_out/r4/ns_ab.txt
done; done
