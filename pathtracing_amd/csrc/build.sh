#!/bin/sh
# Builds the two product libraries in-tree:
#   pathtracing_amd/libptkhost.so  host loader (g++)
#   pathtracing_amd/libptk.so      HIP kernels + C ABI for gfx950 (hipcc cross-compiles without a GPU)
set -e
cd "$(dirname "$0")"
OUT=..
g++ -std=c++17 -O2 -Wall -Wextra -fPIC -shared -I../../include host/host_loader.cpp -o $OUT/libptkhost.so
# -ffp-contract=off: no fused multiply-adds the source does not ask for.  Every kernel then performs the same
# IEEE operations for the same expression (results do not depend on which kernel variant shaded a vertex) and the
# same ones as the CPU build of the reference (x86-64 without FMA contraction); costs about 4 % on the Cornell scene.
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Idevice -I. ${PTK_HIPFLAGS}"
mkdir -p build
# The instantiations of the path-tracing kernel templates are spread over device/variants/*.hip so that they
# compile side by side (PTK_JOBS at a time).
SOURCES="device/kernels.hip device/denoise.hip device/bvh_build.hip ptk_api.hip $(ls device/variants/*.hip)"
OBJECTS=""
for src in $SOURCES; do OBJECTS="$OBJECTS build/$(basename $src .hip).o"; done
g++ -std=c++17 -O3 -fPIC -I../../include -I. -c host/bvh_builder.cpp -o build/bvh_builder.o &
printf '%s\n' $SOURCES | xargs -P "${PTK_JOBS:-8}" -I{} sh -c "hipcc $HIPFLAGS -c {} -o build/\$(basename {} .hip).o"
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OBJECTS build/bvh_builder.o -o $OUT/libptk.so -lpthread
mkdir -p $OUT/bin
for app in pathtracer denoiser; do
  g++ -std=c++17 -O2 -Wall -Wextra -pthread -I../../include app/${app}_main.cpp -o $OUT/bin/$app -L$OUT -lptk -lptkhost \
      -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath-link,/opt/rocm/lib
done
echo "built $OUT/libptkhost.so $OUT/libptk.so $OUT/bin/pathtracer $OUT/bin/denoiser"
