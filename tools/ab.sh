timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for flow in 1 0; do
  echo "== FLOW=$flow N=1"; DPGO_SPD_FLOW=$flow timeout 300 python bench.py --no-cpu --no-prof | cut -c100-260
  echo "== FLOW=$flow emu8"; DPGO_SPD_FLOW=$flow timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof | cut -c100-260
done
