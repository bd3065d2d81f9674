mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r4/last_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r4/last_tests.txt 2>&1
timeout 600 python bench.py > gpurun_out/r4/last_bench.json 2> gpurun_out/r4/last_bench.err
