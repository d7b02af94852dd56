#!/bin/bash
# One rocprofv3 counter pass over a short bench run (GPU box).  usage: tools/pmc_pass.sh <tag> "<counters>" [bench flags...]
# PTK_DEVICE_LIBRARY in the environment selects the device library build, as for bench.py itself.
set -e
tag=$1; counters=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc $counters -d "$out" -o pmc --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --only-headline --steps 1 --warmup 1 "$@" > "$out/bench.log" 2>&1
