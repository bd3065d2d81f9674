# on the GPU box: for every .ab/lib_*.so run the two benches in both solve modes
cp dpgo_amd/libdpgo_amd.so /tmp/lib_keep.so
for lib in .ab/lib_*.so; do
  cp $lib dpgo_amd/libdpgo_amd.so
  for flow in 0 1; do
    a=$(DPGO_SPD_FLOW=$flow timeout 300 python bench.py --no-cpu --no-prof 2>/dev/null | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    b=$(DPGO_SPD_FLOW=$flow timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof 2>/dev/null | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$lib flow=$flow  n1 $a ms   emu8 $b ms"
  done
done
cp /tmp/lib_keep.so dpgo_amd/libdpgo_amd.so
