#!/bin/bash
# The host code under sanitizers, on a CPU build (no GPU, no HIP):
#   thread     tests/cpp/bvh_check.cpp + host/bvh_builder.cpp with -fsanitize=thread: the multi-threaded hierarchy build
#              on uniform and lopsided meshes (the shapes of tests/test_bvh_builder.py), eight build threads; and
#              tests/cpp/load_scene_main.cpp + host/host_loader.cpp: the OBJ reader parsing a mesh in eight parts
#   address    tests/cpp/load_scene_main.cpp + host/host_loader.cpp with -fsanitize=address,undefined: tools/fuzz_loader.py
#              feeds it mutated scene directories; every load must end with "loaded" or "error:", never a report
# usage: tools/sanitize_host.sh [thread|address|all] [first_seed last_seed]      (default: all 0 60)
# Builds into a temporary directory; exit status 0 when no sanitizer reported anything.
set -e
what=${1:-all}
first=${2:-0}
last=${3:-60}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$(mktemp -d)
trap 'rm -rf "$out"' EXIT
inc="-I$root/include -I$root/pathtracing_amd/csrc"
if [ "$what" = thread ] || [ "$what" = all ]; then
  g++ -std=c++17 -O1 -g -fsanitize=thread -DPTK_BVH_CHECK_THREADS=8 $inc "$root/tests/cpp/bvh_check.cpp" \
      "$root/pathtracing_amd/csrc/host/bvh_builder.cpp" -o "$out/bvh_check_tsan" -lpthread
  for args in "20000 2 4" "60000 2 5 clustered 64" "60000 2 5 clustered 20" "50000 1 3 sliver 64" "50000 1 3 sliver 30"; do
    TSAN_OPTIONS="halt_on_error=1 exitcode=66" "$out/bvh_check_tsan" $args
  done
  # the OBJ reader cuts its file into parts that are parsed on threads of their own: a generated mesh in eight parts
  g++ -std=c++17 -O1 -g -fsanitize=thread $inc "$root/tests/cpp/load_scene_main.cpp" "$root/pathtracing_amd/csrc/host/host_loader.cpp" \
      -o "$out/load_scene_tsan" -lpthread
  python3 -c "import sys; sys.path.insert(0, '$root'); from pathtracing_amd import scenes; scenes.mesh_box('$out/mesh', n_lon=96, n_lat=49, material='gold', WIDTH=16, HEIGHT=16)"
  PTK_OBJ_PARSE_PARTS=8 TSAN_OPTIONS="halt_on_error=1 exitcode=66" "$out/load_scene_tsan" "$out/mesh"
  echo "thread sanitizer: no report"
fi
if [ "$what" = address ] || [ "$what" = all ]; then
  g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -static-libasan -static-libubsan $inc \
      "$root/tests/cpp/load_scene_main.cpp" "$root/pathtracing_amd/csrc/host/host_loader.cpp" -o "$out/load_scene_asan"
  PTK_FUZZ_LOADER_EXE="$out/load_scene_asan" ASAN_OPTIONS="detect_leaks=1 exitcode=67" python3 "$root/tools/fuzz_loader.py" "$first" "$last" > "$out/fuzz.log" 2>&1 || {
    tail -40 "$out/fuzz.log"
    exit 1
  }
  tail -1 "$out/fuzz.log"
  echo "address and undefined-behaviour sanitizers: no report"
fi
