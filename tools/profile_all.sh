#!/bin/bash
# Runs on the GPU box (through gpurun): tools/profile_round.sh for every workload of the round, then the denoise
# kernels under the kernel trace.  Output under gpurun_out/<round>_<workload>/; digest afterwards, in the build
# container, with tools/profile_digest_all.sh <round>.   usage: tools/profile_all.sh <round>
set -e
rnd=$1
R=${GRAFT_REPO_ROOT:-$PWD}
for w in example mesh room cloud; do
  steps=16; [ "$w" = example ] || steps=8
  # one batch at a time (batch_lanes=1): with two in flight the kernels of the two share the device and their durations say less
  "$R/tools/profile_round.sh" "${rnd}_$w" --workload $w --steps $steps --option batch_lanes=1
done
mkdir -p "$R/gpurun_out/${rnd}_denoise"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/${rnd}_denoise/trace" -o trace --output-format csv -- python3 "$R/tools/denoise_bench.py" 10 > "$R/gpurun_out/${rnd}_denoise/bench.log" 2>&1
cat "$R/gpurun_out/${rnd}_denoise/bench.log" | grep "ms per"
