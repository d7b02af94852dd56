#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the default bench run, then
# the HBM traffic counters in their own passes.  Output under gpurun_out/<tag>/.
# usage: tools/profile_round.sh <tag> [bench flags...]
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline "$@" > "$out/bench_trace.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --steps 2 --warmup 1 "$@" > "$out/bench_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --steps 2 --warmup 1 "$@" > "$out/bench_write.log" 2>&1
tail -1 "$out/bench_trace.log"
