#!/bin/bash
# Runs bench.py once per device-library build given as argument and prints value + family times.
# usage: tools/ab_bench.sh <lib.so> [<lib.so> ...]   (extra bench flags via BENCH_FLAGS)
set -e
mkdir -p gpurun_out/ab
for lib in "$@"; do
  name=$(basename "$lib" .so)
  PTK_DEVICE_LIBRARY="$PWD/$lib" timeout -k 10 300 python bench.py --no-cpu-baseline $BENCH_FLAGS > "gpurun_out/ab/$name.log" 2>&1
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab/{name}.log").read().strip().splitlines()[-1])
print(name, round(d["value"], 1), {k: round(v, 1) for k, v in d["roofline"]["device_ms"].items()})
PY
done
