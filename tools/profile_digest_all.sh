#!/bin/bash
# Turns the output of tools/profile_all.sh into the files kept under profiles/.  usage: tools/profile_digest_all.sh <round>
set -e
rnd=$1
cd "$(dirname "$0")/.."
for w in example mesh room cloud denoise; do
  python3 tools/profile_digest.py gpurun_out/${rnd}_$w $rnd $w
  python3 tools/counters_digest.py $rnd $w gpurun_out/${rnd}_$w/sq_a/pmc_counter_collection.csv gpurun_out/${rnd}_$w/sq_b/pmc_counter_collection.csv gpurun_out/${rnd}_$w/sq_c/pmc_counter_collection.csv
done
grep "ms per" gpurun_out/${rnd}_denoise/bench_trace.log > profiles/${rnd}_denoise_times.txt
