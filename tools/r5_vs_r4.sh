# same-box, interleaved: the round-4 tree (.ab/r4tree, its own library and python package) against the current one on the
# parity configurations (the 40 / 60 / 200-iteration windows of tests/config_rates.py) and on the emulated rank.
# Before sending the repo to the GPU box:  git worktree add .ab/r4tree 93284b7 && make -C .ab/r4tree/dpgo_amd/csrc -j8
# (and let its tests/config_rates.py skip the oracle when NO_ORACLE is set); afterwards: git worktree remove --force .ab/r4tree
tag=${1:-r5/vs_r4}; mkdir -p gpurun_out/$tag
for rep in 1 2 3 4; do
  echo "== rep $rep: round 4"
  (cd .ab/r4tree && NO_ORACLE=1 timeout 600 python tests/config_rates.py 2>&1 >/dev/null | grep config | sed 's/oracle.*//')
  (cd .ab/r4tree && timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('emulated rank 3 of 8: %.4f ms / iteration' % j['ms_per_step'])")
  for g in auto 0 1; do
    [ $g = auto ] && unset DPGO_ITER_GRAPH || export DPGO_ITER_GRAPH=$g
    echo "== rep $rep: round 5, DPGO_ITER_GRAPH=$g"
    timeout 600 python tests/config_rates.py --no-oracle 2>&1 >/dev/null | grep config | sed 's/oracle.*//'
    timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --converge 0 --steps 60 --warmup 10 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('emulated rank 3 of 8: %.4f ms / iteration' % j['ms_per_step'], j['graphs'])"
  done
  unset DPGO_ITER_GRAPH
done 2>&1 | tee gpurun_out/$tag/summary.txt
