#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc counter_collection CSVs (one directory per pass).

usage: tools/pmc_summary.py gpurun_out/pmc_sq/sq_counter_collection.csv [more.csv ...]
"""
import collections
import csv
import re
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dispatches = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("ptk::", "")[:28]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in dispatches[k]:
            dispatches[k].add(r["Dispatch_Id"])
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, dispatches, dur


def main():
    for path in sys.argv[1:]:
        agg, dispatches, dur = load(path)
        print("==", path)
        for k, c in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
            parts = [f"{k:28s} n={len(dispatches[k]):5d} ms={dur[k] / 1e6:8.2f}"]
            w = c.get("SQ_WAVES", 0)
            for name, v in sorted(c.items()):
                if name == "SQ_WAVES":
                    parts.append(f"waves={v:.0f}")
                elif name.startswith("SQ_INSTS") and w:
                    parts.append(f"{name[9:]}/wave={v / w:.0f}")
                elif name in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY",
                              "SQ_INST_CYCLES_VMEM") and c.get("SQ_WAVE_CYCLES"):
                    parts.append(f"{name[3:]}={100 * v / c['SQ_WAVE_CYCLES']:.1f}%")
                elif name == "SQ_WAVE_CYCLES" and w:
                    parts.append(f"cyc/wave={4 * v / w:.0f}")
                elif name in ("FETCH_SIZE", "WRITE_SIZE"):
                    parts.append(f"{name}={v / 1e6:.2f} GB")
                else:
                    parts.append(f"{name}={v:.3g}")
            print(" ".join(parts))


if __name__ == "__main__":
    main()
