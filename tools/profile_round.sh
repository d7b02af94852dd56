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
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --only-headline "$@" > "$out/bench_trace.log" 2>&1
rm -f "$out"/trace/*_kernel_trace.csv   # one row per launch: tens of megabytes; the stats file is what is kept
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --only-headline --steps 2 --warmup 1 --spp-per-step 8 "$@" > "$out/bench_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --only-headline --steps 2 --warmup 1 --spp-per-step 8 "$@" > "$out/bench_write.log" 2>&1
# the SQ counters, two passes of at most eight
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY -d "$out/sq_a" -o pmc --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --only-headline --steps 1 --warmup 1 --spp-per-step 8 "$@" > "$out/bench_sq_a.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS -d "$out/sq_b" -o pmc --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --only-headline --steps 1 --warmup 1 --spp-per-step 8 "$@" > "$out/bench_sq_b.log" 2>&1
# what the waves wait for: any wait, any issue, the memory pipes, and the clock the device ran at
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d "$out/sq_c" -o pmc --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline --only-headline --steps 1 --warmup 1 --spp-per-step 8 "$@" > "$out/bench_sq_c.log" 2>&1
tail -1 "$out/bench_trace.log"
