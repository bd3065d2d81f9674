#!/bin/bash
# Host code (loader, partition, assembly, multifrontal factorisation + host solve, chordal init, exchange plan, C ABI) under
# AddressSanitizer: CPU build only (GPU ASan is not available on this pool).  Usage: bash tools/asan_host_tests.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/dpgo_asan
mkdir -p $out
cd $root/dpgo_amd/csrc
for f in graph spd assemble chordal group tnt dchordal comm capi; do
  hipcc --offload-arch=gfx950 -std=c++17 -O1 -g -fPIC -fopenmp -fsanitize=address -fno-omit-frame-pointer -Wno-option-ignored -c $f.cpp -o $out/$f.o
done
hipcc --offload-arch=gfx950 -std=c++17 -O1 -fPIC -fopenmp -c kernels.hip -o $out/kernels.o
hipcc --offload-arch=gfx950 -std=c++17 -O1 -fPIC -fopenmp -c spd_dev.hip -o $out/spd_dev.o
hipcc --offload-arch=gfx950 -shared -fopenmp -fsanitize=address -shared-libsan -o $out/libdpgo_amd.so $out/*.o -ldl
rt=$(ldd $out/libdpgo_amd.so | awk '/asan/ {print $3}')
cp $root/dpgo_amd/libdpgo_amd.so $out/keep.so
trap 'cp $out/keep.so $root/dpgo_amd/libdpgo_amd.so' EXIT
cp $out/libdpgo_amd.so $root/dpgo_amd/libdpgo_amd.so
cd $root
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$rt python -m pytest tests/test_host_logic.py tests/test_exchange_gloo.py -x -q
