# Round artefacts (run on the GPU box): GPU tests, default bench line, emulated 8-GPU rank, rocprofv3 kernel
# stats and the two PMC passes.  Usage: bash tools/final_profile.sh r01
tag=${1:-r04}
out=gpurun_out/final
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $out/gpu_tests.txt
timeout 600 python bench.py > $out/${tag}_bench_n1.json 2> $out/bench_n1.err
timeout 600 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu > $out/${tag}_bench_emulated_rank3of8.json 2> $out/bench_emu.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu --no-prof --converge 0 > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_bench_default.csv
python3 tools/trace_tail.py $out/stats 110 > $out/${tag}_timeline_last_step_n1.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 bench.py --no-cpu --no-prof --steps 5 --warmup 2 --converge 0 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 bench.py --no-cpu --no-prof --steps 5 --warmup 2 --converge 0 > $out/pmc_write.log 2>&1
python3 tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/${tag}_pmc_hbm_traffic.json "50,50,40,400000" 1 > $out/pmc_summary.txt 2>&1
# the MFMA kernel of the device factorisation: its rate from HIP events (DPGO_SPD_DUMP) and the matrix-pipe busy cycles
DPGO_SPD_DUMP=1 python3 bench.py --no-cpu --no-prof --converge 0 --steps 3 --warmup 1 2>&1 >/dev/null | grep "device factorisation" > $out/${tag}_mfma_factor_rate.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $out/pmc_mfma -- python3 bench.py --no-cpu --no-prof --converge 0 --steps 3 --warmup 1 > $out/pmc_mfma.log 2>&1
python3 tools/mfma_summary.py $out/pmc_mfma $out/${tag}_mfma_factor_rate.txt $out/${tag}_mfma_utilisation.json > $out/mfma_summary.txt 2>&1
rm -rf $out/stats $out/pmc_fetch $out/pmc_write $out/pmc_mfma
cat $out/gpu_tests.txt; cut -c1-300 $out/${tag}_bench_n1.json; cat $out/pmc_summary.txt; cat $out/mfma_summary.txt
