# Round artefacts, on the GPU box: bash tools/final_profile.sh <tag> [stage ...]   (default: every stage; outputs under
# gpurun_out/final, to be copied into profiles/).  Stages:
#   bench    the default bench line (roofline, cpu_baseline, convergence), the emulated rank's line (with PMC traffic)
#   stats    rocprofv3 --kernel-trace --stats of the default command, the last step's timeline, the two PMC passes
#   mfma     the factorisation's MFMA kernel: rate and matrix-pipe counters
#   levels   per-level solve tables (N = 1 and one node per GPU), the one-node timeline
#   dynamic  Dynamic-rescale probe and the launches of one refactorisation
#   rates    parity-configuration rates (5 repetitions, median), every emulated rank of the 8 / 4 / 2-GPU splits
#   host     HIP API calls per iteration (eager / replayed), the exchange against itself, round 6's launch sequence against round 5's
#   cpu      the CPU side of the convergence metric (5 minutes of box time)
tag=${1:-r06}; shift
stages="${*:-bench stats mfma levels dynamic rates host cpu}"
out=gpurun_out/final
mkdir -p $out
has() { case " $stages " in *" $1 "*) return 0;; esac; return 1; }
prof() { ( cd /tmp && TMPDIR=/tmp "$@" ); }
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
if has bench; then
  timeout 900 python bench.py > $out/${tag}_bench_n1.json 2> $out/bench_n1.err
  timeout 600 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu > $out/${tag}_bench_emulated_rank3of8.json 2> $out/bench_emu.err
  cut -c1-400 $out/${tag}_bench_n1.json
fi
if has stats; then
  prof rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -- python3 $R/bench.py --no-cpu --no-prof --converge 0 --traffic off > $out/stats.log 2>&1
  cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_bench_default.csv
  python3 tools/trace_tail.py $out/stats 110 > $out/${tag}_timeline_last_step_n1.txt
  prof rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/pmc_fetch -- python3 $R/bench.py --no-cpu --no-prof --steps 5 --warmup 2 --converge 0 --traffic off > $out/pmc_fetch.log 2>&1
  prof rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/pmc_write -- python3 $R/bench.py --no-cpu --no-prof --steps 5 --warmup 2 --converge 0 --traffic off > $out/pmc_write.log 2>&1
  python3 tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/${tag}_pmc_hbm_traffic.json "50,50,40,400000" 1 > $out/pmc_summary.txt 2>&1
  rm -rf $out/stats $out/pmc_fetch $out/pmc_write; cat $out/pmc_summary.txt
fi
if has mfma; then
  DPGO_SPD_DUMP=1 python3 bench.py --no-cpu --no-prof --converge 0 --traffic off --steps 3 --warmup 1 2>&1 >/dev/null | grep "device factorisation" > $out/${tag}_mfma_factor_rate.txt
  prof rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $R/$out/pmc_mfma -- python3 $R/bench.py --no-cpu --no-prof --converge 0 --traffic off --steps 3 --warmup 1 > $out/pmc_mfma.log 2>&1
  python3 tools/mfma_summary.py $out/pmc_mfma $out/${tag}_mfma_factor_rate.txt $out/${tag}_mfma_utilisation.json > $out/mfma_summary.txt 2>&1
  rm -rf $out/pmc_mfma; cat $out/mfma_summary.txt
fi
if has levels; then
  bash tools/spd_profile.sh $tag
  cp gpurun_out/spd_${tag}_n1.txt $out/${tag}_spd_levels.txt
  cp gpurun_out/spd_${tag}_emu8.txt $out/${tag}_spd_levels_one_node.txt
  bash tools/trace_levels.sh one_node --emulate-world 8 --emulate-rank 3
  tail -60 gpurun_out/timeline_one_node.txt > $out/${tag}_timeline_last_step_one_node.txt
fi
if has dynamic; then
  python3 tools/probes/dynamic_headline.py 50,50,40,400000 40 > $out/${tag}_dynamic_headline.txt 2>&1
  prof rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/dyntr -- python3 $R/tools/probes/dynamic_headline.py 50,50,40,400000 12 > gpurun_out/dyntr.log 2>&1
  python3 tools/probes/dyn_trace.py gpurun_out/dyntr > $out/${tag}_dynamic_refactorisation_launches.txt
  rm -rf gpurun_out/dyntr; tail -3 $out/${tag}_dynamic_headline.txt
fi
if has rates; then
  for rep in 1 2 3 4 5; do python tests/config_rates.py --repeat 5 $( [ $rep -gt 1 ] && echo --no-oracle ) > $out/rates_$rep.json 2>> $out/config_rates.err; done
  python3 - > $out/${tag}_config_rates.json <<PY
import json, statistics
runs = [json.load(open("$out/rates_%d.json" % r)) for r in range(1, 6)]
res = []
for i, c in enumerate(runs[0]):
    g = [r[i]["gpu_iters_per_s"] for r in runs]
    res.append(dict(c, gpu_iters_per_s=statistics.median(g), gpu_iters_per_s_runs=g, note="GPU: median of 5 runs on one box, each over 5 x the oracle's window after 3 warm-up iterations (DPGO_ITER_GRAPH unset: replays by measurement); oracle: one run of its window from the same initial point"))
print(json.dumps(res))
PY
  rm -f $out/rates_?.json
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
  cat $out/${tag}_emulated_all_ranks.txt | head -12
fi
if has host; then
  bash tools/r5_api.sh final; cat gpurun_out/final/api_emulated_rank_graph0.txt gpurun_out/final/api_emulated_rank_graph1.txt > $out/${tag}_api_calls_per_iteration.txt
  bash tools/r6_xchg.sh final "DPGO_X=0" > /dev/null 2>&1; cp gpurun_out/final/self_exchange.txt $out/${tag}_self_exchange_runs.txt
  # per-iteration view of the emulated rank's kernels: round 6's launch sequence and round 5's (DPGO_FUSED=0), same box
  bash tools/r6_prof.sh final 60 "DPGO_X=0" "DPGO_FUSED=0"
  { cat gpurun_out/final/iter_1.txt; echo; cat gpurun_out/final/iter_2.txt; } > $out/${tag}_kernels_per_iteration_one_node.txt
  # the early regime, five windows per run, three runs each, interleaved: round 6's sequence against round 5's
  bash tools/r6_windows.sh --emulate-world 8 --emulate-rank 3 -- "DPGO_X=0" "DPGO_FUSED=0" > $out/${tag}_fused_vs_unfused_one_node.txt 2>&1
  bash tools/r6_windows.sh -- "DPGO_X=0" "DPGO_FUSED=0" > $out/${tag}_fused_vs_unfused_n1.txt 2>&1
fi
if has cpu; then
  python tools/cpu_convergence.py > $out/${tag}_cpu_convergence.json 2> $out/cpu_conv.err
fi
ls $out
