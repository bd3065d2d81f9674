#!/bin/bash
# build_variant.sh NAME [-DX=Y ...]: A/B builds of the kernels with different defines -> .ab/lib_NAME.so
set -e
name=$1; shift
cd "$(dirname "$0")/../dpgo_amd/csrc"
mkdir -p ../../.ab
hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -fopenmp -Wno-unused-function "$@" -c kernels.hip -o /tmp/kernels_$name.o
hipcc --offload-arch=gfx950 -shared -fopenmp -o ../../.ab/lib_$name.so graph.o spd.o assemble.o chordal.o group.o tnt.o dchordal.o comm.o capi.o spd_dev.o /tmp/kernels_$name.o -ldl
