for c in 0 1 2 3; do
  a=$(DPGO_SPD_COLLAPSE=$c timeout 300 python bench.py --no-cpu --no-prof 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.3f' % j['ms_per_step'], j['solver'])")
  b=$(DPGO_SPD_COLLAPSE=$c timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.3f' % j['ms_per_step'], j['solver'])")
  echo "collapse=$c  n1 $a | emu8 $b"
done
