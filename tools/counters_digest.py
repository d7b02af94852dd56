#!/usr/bin/env python3
"""Turns rocprofv3 --pmc passes (tools/pmc_pass.sh) into profiles/<round>_<workload>_counters.md and
profiles/counters_latest.json: per kernel, VALU instructions per wave, the share of the SIMDs' time spent issuing
them and the occupancy the kernel ran at.

usage: tools/counters_digest.py <round> <workload> <pass.csv> [<pass.csv> ...]
Counters used (any subset): SQ_WAVES, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_ACTIVE_INST_VALU,
SQ_WAIT_INST_ANY, SQ_THREAD_CYCLES_VALU, SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64, SQ_INSTS_VALU_INT32, SQ_INSTS_VALU_INT64,
GRBM_GUI_ACTIVE.  SQ_WAVE_CYCLES, SQ_ACTIVE_INST_VALU and SQ_THREAD_CYCLES_VALU count in units of four cycles.
"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 256 * 4  # MI355X
sys.path.insert(0, ROOT)
from bench import sources_hash  # noqa: E402


def main():
    rnd, workload, paths = sys.argv[1], sys.argv[2], sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    seen = collections.defaultdict(set)
    per_pass = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    pass_dur = collections.defaultdict(lambda: collections.defaultdict(float))
    first_pass_of = {}
    for path in paths:
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").strip()
            if k.startswith("__amd"):
                continue
            per_pass[k][path][r["Counter_Name"]] += float(r["Counter_Value"])
            if first_pass_of.setdefault(r["Counter_Name"], path) == path:  # a counter collected in two passes counts once
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (path, r["Dispatch_Id"])
            if key not in seen[k]:
                seen[k].add(key)
                pass_dur[k][path] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                if path == paths[0]:
                    dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows, latest = [], {}
    for k, c in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
        waves = c.get("SQ_WAVES", 0)
        if not waves:
            continue
        wave_cycles = c.get("SQ_WAVE_CYCLES", 0)   # x4 cycles, summed over waves
        entry = {"waves": waves, "launches": len([1 for key in seen[k] if key[0] == paths[0]])}
        if c.get("SQ_INSTS_VALU"):
            entry["valu_per_wave"] = c["SQ_INSTS_VALU"] / waves
        if c.get("SQ_INSTS_SALU"):
            entry["salu_per_wave"] = c["SQ_INSTS_SALU"] / waves
        if wave_cycles and c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_BUSY_CYCLES"):
            # average resident waves per SIMD = wave-cycles / (cycles the kernel was resident x SIMDs)
            busy = c["SQ_BUSY_CYCLES"] / 4.0  # reported per shader engine quadrant in cycles; see the table note
            entry["valu_share_of_wave_time"] = c["SQ_ACTIVE_INST_VALU"] / wave_cycles
            # over the kernel's own duration in this pass (timestamps in ns, 2.4 GHz): resident waves per SIMD, and the
            # share of the SIMDs' issue slots spent on vector instructions
            if dur[k] > 0:
                simd_cycles = dur[k] * 2.4 * SIMDS
                entry["waves_per_simd"] = 4.0 * wave_cycles / simd_cycles
                entry["simd_valu_busy"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles
        if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
            entry["active_lanes_per_valu"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
        if wave_cycles and c.get("SQ_WAIT_INST_ANY"):
            entry["wait_share_of_wave_time"] = c["SQ_WAIT_INST_ANY"] / wave_cycles
        # the third pass has its own SQ_WAVE_CYCLES (agg sums passes, so use the per-pass figures kept below)
        third = per_pass[k][paths[-1]] if len(paths) > 2 else {}
        if third.get("SQ_WAVE_CYCLES"):
            for name, key in (("SQ_WAIT_ANY", "any_wait_share"), ("SQ_ACTIVE_INST_ANY", "issue_share"),
                              ("SQ_ACTIVE_INST_VMEM", "vmem_issue_share"), ("SQ_ACTIVE_INST_LDS", "lds_issue_share"),
                              ("SQ_ACTIVE_INST_SCA", "scalar_issue_share")):
                if name in third:
                    entry[key] = third[name] / third["SQ_WAVE_CYCLES"]
            if third.get("GRBM_GUI_ACTIVE") and pass_dur[k].get(paths[-1]):
                entry["clock_ghz"] = third["GRBM_GUI_ACTIVE"] / 8.0 / pass_dur[k][paths[-1]]  # summed over the 8 XCDs; cycles per nanosecond
        for name in ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "INT32", "INT64"):
            v = c.get("SQ_INSTS_VALU_" + name)
            if v:
                entry[name.lower() + "_per_wave"] = v / waves
        latest[k] = entry
        rows.append((k, entry))
    out_dir = os.path.join(ROOT, "profiles")
    js = os.path.join(out_dir, "counters_latest.json")
    data = json.load(open(js)) if os.path.exists(js) else {}
    data[workload] = latest
    data.setdefault("_sources", {})[workload] = sources_hash()  # the code these numbers were measured on (bench.py checks it)
    json.dump(data, open(js, "w"), indent=1, sort_keys=True)
    with open(os.path.join(out_dir, f"{rnd}_{workload}_counters.md"), "w") as f:
        f.write(f"# SQ counters per kernel, workload `{workload}` ({rnd})\n\n")
        command = "tools/denoise_bench.py 2" if workload == "denoise" else "bench.py --steps 1 --warmup 1 --spp-per-step 8"
        f.write(f"rocprofv3 `--pmc`, one pass per counter group (`tools/profile_round.sh`, `tools/profile_all.sh`), `{command}`.\n")
        f.write("`VALU share` = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: the part of a wave's life spent issuing vector\n")
        f.write("instructions; times the waves per SIMD the kernel is compiled for it is the share of the SIMD's issue\n")
        f.write("slots in use.  `lanes` = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU: active lanes per vector instruction.\n")
        f.write("`any wait` = SQ_WAIT_ANY / SQ_WAVE_CYCLES, `issuing` = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES and the per-pipe issue\n")
        f.write("shares come from a third pass; `clock` = GRBM_GUI_ACTIVE over the kernel's duration in that pass.\n\n")
        f.write("| kernel | waves | VALU/wave | SALU/wave | VALU share | wait share | lanes | fma/mul/add/trans f64, int32, int64 per wave | any wait | issuing | vmem / lds / scalar issue | clock GHz |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k, e in rows:
            mix = "/".join(f"{e.get(n + '_per_wave', 0):.0f}" for n in ("fma_f64", "mul_f64", "add_f64", "trans_f64")) + \
                  f", {e.get('int32_per_wave', 0):.0f}, {e.get('int64_per_wave', 0):.0f}"
            f.write(f"| `{k}` | {e['waves']:.0f} | {e.get('valu_per_wave', 0):.0f} | {e.get('salu_per_wave', 0):.0f} | "
                    f"{100 * e.get('valu_share_of_wave_time', 0):.1f} % | {100 * e.get('wait_share_of_wave_time', 0):.1f} % | "
                    f"{e.get('active_lanes_per_valu', 0):.1f} | {mix} | {100 * e.get('any_wait_share', 0):.1f} % | "
                    f"{100 * e.get('issue_share', 0):.1f} % | {100 * e.get('vmem_issue_share', 0):.1f} / "
                    f"{100 * e.get('lds_issue_share', 0):.1f} / {100 * e.get('scalar_issue_share', 0):.1f} % | {e.get('clock_ghz', 0):.2f} |\n")
    print("wrote", out_dir)


if __name__ == "__main__":
    main()
