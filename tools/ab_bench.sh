#!/bin/bash
# A/B of device-library builds on the GPU box.  usage: tools/ab_bench.sh <workload> <tag>...
# Each tag names a build kept beside the product library as pathtracing_amd/libptk_<tag>.so (built here with
# PTK_HIPFLAGS=... sh pathtracing_amd/csrc/build.sh and copied); bench.py loads it through PTK_DEVICE_LIBRARY.
# Prints Mpaths/s, ms per step and the device time by kernel group of every build; a build compiled with
# -DPTK_TRAVERSE_PROBE also prints k_traverse's pass and lane counts.
wl=$1; shift
for tag in "$@"; do
  PTK_DEVICE_LIBRARY=$PWD/pathtracing_amd/libptk_$tag.so timeout -k 10 150 python bench.py --only-headline --no-cpu-baseline --workload $wl --steps 8 --warmup 2 > gpurun_out/ab_$tag.log 2> gpurun_out/ab_$tag.err || exit 1
  grep "traverse probe" gpurun_out/ab_$tag.err | tail -1
  python -c "
import json;d=json.loads(open('gpurun_out/ab_$tag.log').read().strip().splitlines()[-1]);print('$wl $tag', round(d['value'],1),round(d['ms_per_step'],3),{k:round(v,2) for k,v in d['roofline']['device_ms'].items()})"
done
