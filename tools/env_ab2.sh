# usage: env_ab2.sh "A=1 B=2" "A=3 B=4" ...  -> headline and emulated-rank ms/step for each environment
for v in "$@"; do
  a=$(env $v timeout 300 python bench.py --no-cpu --no-prof 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.3f' % j['ms_per_step'])")
  b=$(env $v timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.3f' % j['ms_per_step'])")
  echo "$v  n1 $a | emu8 $b"
done
