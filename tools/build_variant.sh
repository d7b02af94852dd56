#!/bin/bash
# Builds pathtracing_amd/libptk_<name>.so with extra compiler flags (tuning / profiling builds).
# usage: tools/build_variant.sh <name> <flags...>     e.g. tools/build_variant.sh stamps -DPTK_STAMPS
set -e
name=$1; shift
cd "$(dirname "$0")/../pathtracing_amd/csrc"
tmp=$(mktemp -d)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Idevice -I. $*"
hipcc $F -c device/kernels.hip -o $tmp/kernels.o &
hipcc $F -c ptk_api.hip -o $tmp/ptk_api.o &
hipcc $F -c device/denoise.hip -o $tmp/denoise.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC $tmp/kernels.o $tmp/denoise.o $tmp/ptk_api.o build/bvh_builder.o -o ../libptk_$name.so -lpthread
rm -rf $tmp
echo built pathtracing_amd/libptk_$name.so
