#!/bin/bash
# Prints registers, spills, scratch and LDS of every kernel in the object files of the device library
# (pathtracing_amd/csrc/build/*.o after build.sh, or the directory given), read from the metadata notes of their
# gfx950 code objects.  usage: tools/kernel_resources.sh [name filter regex] [object directory]
set -e
filter=${1:-.}
dir=$(realpath "${2:-$(dirname "$0")/../pathtracing_amd/csrc/build}")
tmp=$(mktemp -d)
cd "$tmp"
for o in "$dir"/*.o; do
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=fat.bin "$o" 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fat.bin --output=code.co --unbundle 2>/dev/null || continue
  [ -s code.co ] || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes code.co >> notes.txt 2>/dev/null || true
done
python3 - "$filter" <<'PY'
import re, subprocess, sys
text = open("notes.txt").read()
flt = re.compile(sys.argv[1])
rows = []
for block in text.split("  - .agpr_count:")[1:]:
    get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", block) or [None, "?"])[1]
    name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))
    if flt.search(name):
        rows.append(f"{name:60s} vgpr {get('vgpr_count'):>4} spill {get('vgpr_spill_count'):>4} sgpr {get('sgpr_count'):>4} sspill {get('sgpr_spill_count'):>4} "
                    f"scratch {get('private_segment_fixed_size'):>5} lds {get('group_segment_fixed_size'):>6}")
print("\n".join(sorted(rows)))
PY
rm -rf "$tmp"
