#!/bin/bash
# Runs on the GPU box (through gpurun), after tools/profile_digest_all.sh has refreshed profiles/*_latest.json: the
# default bench line, then one line per further workload, each with the roofline figures of the current sources.
# Output: gpurun_out/<round>_records/<name>.json.   usage: tools/bench_records.sh <round>
set -e
rnd=$1
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/${rnd}_records
mkdir -p "$out"
cd "$R"
timeout -k 10 400 python bench.py > "$out/default.json"
for w in mesh room cloud; do
  timeout -k 10 200 python bench.py --only-headline --no-cpu-baseline --workload $w --steps 16 > "$out/$w.json"
done
python - "$out" <<'PY'
import json, sys, os
for name in ("default", "mesh", "room", "cloud"):
    d = json.loads(open(os.path.join(sys.argv[1], name + ".json")).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(name, round(d["value"], 1), d["unit"], "| roofline", r["bound"], round(r["frac"], 3), "| stale", r["hbm"].get("source", {}).get("stale"))
PY
