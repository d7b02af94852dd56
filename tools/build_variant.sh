#!/bin/bash
# Builds pathtracing_amd/libptk_<name>.so with extra compiler flags (tuning / profiling builds).
# usage: tools/build_variant.sh <name> <flags...>     e.g. tools/build_variant.sh stamps -DPTK_STAMPS
set -e
name=$1; shift
cd "$(dirname "$0")/../pathtracing_amd/csrc"
tmp=$(mktemp -d)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Idevice -I. $*"
SOURCES="device/kernels.hip device/denoise.hip device/bvh_build.hip ptk_api.hip $(ls device/variants/*.hip)"
printf '%s\n' $SOURCES | xargs -P 8 -I{} sh -c "hipcc $F -c {} -o $tmp/\$(basename {} .hip).o"
DEFINES=$(printf '%s\n' "$@" | grep '^-D' | tr '\n' ' ') # the host-side hierarchy builder sees the same macros
g++ -std=c++17 -O3 -fPIC $DEFINES -I../../include -I. -c host/bvh_builder.cpp -o $tmp/bvh_builder.o
hipcc --offload-arch=gfx950 -shared -fPIC $tmp/*.o -o ../libptk_$name.so -lpthread
rm -rf $tmp
echo built pathtracing_amd/libptk_$name.so
