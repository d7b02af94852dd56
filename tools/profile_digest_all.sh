#!/bin/bash
# Turns the output of tools/profile_all.sh into the files kept under profiles/.  usage: tools/profile_digest_all.sh <round>
set -e
rnd=$1
cd "$(dirname "$0")/.."
for w in example mesh room cloud; do
  python3 tools/profile_digest.py gpurun_out/${rnd}_$w $rnd $w
  python3 tools/counters_digest.py $rnd $w gpurun_out/${rnd}_$w/sq_a/pmc_counter_collection.csv gpurun_out/${rnd}_$w/sq_b/pmc_counter_collection.csv
done
cp gpurun_out/${rnd}_denoise/trace/trace_kernel_stats.csv profiles/${rnd}_denoise_kernel_stats.csv
grep "ms per" gpurun_out/${rnd}_denoise/bench.log > profiles/${rnd}_denoise_times.txt
