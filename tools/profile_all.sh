#!/bin/bash
# Runs on the GPU box (through gpurun): tools/profile_round.sh for every workload of the round, then the denoise
# kernels under the kernel trace.  Output under gpurun_out/<round>_<workload>/; digest afterwards, in the build
# container, with tools/profile_digest_all.sh <round>.   usage: tools/profile_all.sh <round> [workload ...]   (default: example mesh room cloud denoise)
set -e
rnd=$1; shift
wanted=${*:-example mesh room cloud denoise}
R=${GRAFT_REPO_ROOT:-$PWD}
for w in $wanted; do
  [ "$w" = denoise ] && continue
  steps=16; [ "$w" = example ] || steps=8
  # one batch at a time (batch_lanes=1): with two in flight the kernels of the two share the device and their durations say less
  "$R/tools/profile_round.sh" "${rnd}_$w" --workload $w --steps $steps --option batch_lanes=1
done
case " $wanted " in *" denoise "*) ;; *) exit 0;; esac
out=$R/gpurun_out/${rnd}_denoise
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace --output-format csv -- python3 "$R/tools/denoise_bench.py" 10 > "$out/bench_trace.log" 2>&1
rm -f "$out"/trace/*_kernel_trace.csv
grep "ms per" "$out/bench_trace.log"
# the same counters as for the path kernels, over two repetitions of every denoise case
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- python3 "$R/tools/denoise_bench.py" 2 > "$out/bench_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write --output-format csv -- python3 "$R/tools/denoise_bench.py" 2 > "$out/bench_write.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY -d "$out/sq_a" -o pmc --output-format csv -- python3 "$R/tools/denoise_bench.py" 2 > "$out/bench_sq_a.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS -d "$out/sq_b" -o pmc --output-format csv -- python3 "$R/tools/denoise_bench.py" 2 > "$out/bench_sq_b.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d "$out/sq_c" -o pmc --output-format csv -- python3 "$R/tools/denoise_bench.py" 2 > "$out/bench_sq_c.log" 2>&1
