# usage: env_ab.sh VAR v1 v2 ...   -> headline and emulated-rank ms/step for each value of the env var
var=$1; shift
for v in "$@"; do
  a=$(env $var=$v timeout 300 python bench.py --no-cpu --no-prof 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.3f' % j['ms_per_step'])")
  b=$(env $var=$v timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.3f' % j['ms_per_step'])")
  echo "$var=$v  n1 $a | emu8 $b"
done
