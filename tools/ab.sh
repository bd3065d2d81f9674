# on the GPU box: for every .ab/lib_*.so run the headline bench and the emulated 8-GPU rank
cp dpgo_amd/libdpgo_amd.so /tmp/lib_keep.so
for lib in .ab/lib_*.so; do
  cp $lib dpgo_amd/libdpgo_amd.so
  a=$(timeout 300 python bench.py --no-cpu --no-prof 2>/dev/null | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --no-prof 2>/dev/null | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$lib  n1 $a ms   emu8 $b ms"
done
cp /tmp/lib_keep.so dpgo_amd/libdpgo_amd.so
