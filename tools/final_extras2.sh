# Second batch of round artefacts (run on the GPU box after final_profile.sh and final_extras.sh): parity-configuration rates, every emulated rank of the 8 / 4 / 2-GPU splits, the CPU
# side of the convergence metric, the kernel + HIP API trace of AMM-PGO* (no stream synchronisation in star_iterate)
tag=${1:-r04}
out=gpurun_out/final
mkdir -p $out
python tests/config_rates.py > $out/${tag}_config_rates.json 2> $out/config_rates.err
{
echo "# every rank of an N-GPU run emulated on ONE GPU (its nodes only, frozen neighbours, no exchange): ms / iteration"
for r in 0 1 2 3 4 5 6 7; do
  python bench.py --emulate-world 8 --emulate-rank $r --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('8 GPUs, rank $r (node $r): %.4f ms / iteration = %.0f it/s before the exchange' % (j['ms_per_step'], j['value']))"
done
for r in 0 1 2 3; do
  python bench.py --emulate-world 4 --emulate-rank $r --no-cpu --no-prof --converge 0 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('4 GPUs, rank $r: %.4f ms / iteration = %.0f it/s before the exchange' % (j['ms_per_step'], j['value']))"
done
for r in 0 1; do
  python bench.py --emulate-world 2 --emulate-rank $r --no-cpu --no-prof --converge 0 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('2 GPUs, rank $r: %.4f ms / iteration = %.0f it/s before the exchange' % (j['ms_per_step'], j['value']))"
done
} > $out/${tag}_emulated_all_ranks.txt 2>&1
# (SHORT=1: only the rates and the emulated ranks -- the CPU run takes 5 minutes of box time and does not change with the kernels)
[ -n "$SHORT" ] && exit 0
python tools/cpu_convergence.py > $out/${tag}_cpu_convergence.json 2> $out/cpu_conv.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/startr
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d gpurun_out/startr -- python3 tools/probes/star_api_trace.py > gpurun_out/startr.log 2>&1
python3 tools/probes/star_api_summary.py gpurun_out/startr > $out/${tag}_star_api_trace.txt 2>&1
rm -rf gpurun_out/startr
