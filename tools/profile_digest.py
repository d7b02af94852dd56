#!/usr/bin/env python3
"""Turns the output of tools/profile_round.sh into the files kept under profiles/.

usage: tools/profile_digest.py <gpurun_out/tag> <round> <workload>

Writes profiles/<round>_<workload>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim),
profiles/<round>_<workload>_traffic.md and updates profiles/traffic_latest.json with the HBM bytes per launch
of every kernel: FETCH_SIZE and WRITE_SIZE are reported in kilobytes; they were
collected in separate --pmc passes; on gfx950 FETCH_SIZE counts 128-byte requests of wide streaming reads as 64 bytes,
so it is doubled (the records are read 16 bytes per lane).
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, ROOT)
from bench import sources_hash  # noqa: E402


def kernel_name(raw):
    name = re.sub(r"\(.*", "", raw.replace("(anonymous namespace)::", "")).replace("void ", "")
    return name.strip()


def counter_sums(path, counter):
    total = collections.defaultdict(float)
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = kernel_name(r["Kernel_Name"])
        total[k] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return total, {k: len(v) for k, v in launches.items()}


def main():
    src, rnd, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(out_dir, f"{rnd}_{workload}_kernel_stats.csv"))
    with open(os.path.join(src, "bench_trace.log")) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    if lines:
        with open(os.path.join(out_dir, f"{rnd}_{workload}_bench_under_rocprof.json"), "w") as f:
            f.write(lines[-1] + "\n")
    fetch, n_f = counter_sums(os.path.join(src, "fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
    write, n_w = counter_sums(os.path.join(src, "write", "write_counter_collection.csv"), "WRITE_SIZE")
    traffic = {}
    rows = []
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
        if k.startswith("__amd"):
            continue
        n = max(n_f.get(k, 1), 1)
        read_b = 2.0 * fetch[k] * 1000.0 / n       # gfx950: FETCH_SIZE tallies 128-byte requests as 64 bytes
        write_b = write.get(k, 0.0) * 1000.0 / max(n_w.get(k, 1), 1)
        traffic[k] = read_b + write_b
        rows.append((k, n, read_b, write_b))
    latest = os.path.join(out_dir, "traffic_latest.json")
    data = {}
    if os.path.exists(latest):
        with open(latest) as f:
            data = json.load(f)
    data[workload] = traffic
    data.setdefault("_sources", {})[workload] = sources_hash()  # the code these numbers were measured on (bench.py checks it)
    with open(latest, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    with open(os.path.join(out_dir, f"{rnd}_{workload}_traffic.md"), "w") as f:
        f.write(f"# HBM traffic per launch, workload `{workload}` ({rnd})\n\n")
        command = "tools/denoise_bench.py 2" if workload == "denoise" else "bench.py --steps 2 --warmup 1 --spp-per-step 8"
        f.write(f"rocprofv3 `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, one pass each, `{command}`.\n")
        f.write("Counter unit: kilobytes.  Reads = 2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of 16-byte-per-lane\n")
        f.write("streaming reads as 64 bytes); writes = WRITE_SIZE.  Averages over all launches of the kernel in the run.\n\n")
        f.write("| kernel | launches | read MB/launch | write MB/launch | total MB/launch |\n|---|---|---|---|---|\n")
        for k, n, r, w in rows:
            f.write(f"| `{k}` | {n} | {r / 1e6:.2f} | {w / 1e6:.2f} | {(r + w) / 1e6:.2f} |\n")
    print("wrote", out_dir)


if __name__ == "__main__":
    main()
