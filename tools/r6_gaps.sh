cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "DPGO_X=0" "DPGO_FUSE_BEGIN=0"; do
  rm -rf /tmp/pg
  ( export $cfg DPGO_ITER_GRAPH=0; rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -- python3 $R/bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof --traffic off --converge 0 --steps 200 --warmup 10 > /dev/null 2>&1 )
  echo "== $cfg"; python3 $R/tools/trace_gaps.py /tmp/pg 40 | tail -25
done
