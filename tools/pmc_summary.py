#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel (name + grid) from a counter_collection.csv.
usage: python tools/pmc_summary.py <p_counter_collection.csv> [name filter]"""
import collections
import csv
import sys


def main(path, flt=""):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if flt and flt not in name:
            continue
        key = (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        did = (key, r["Dispatch_Id"])
        if did not in seen:
            seen.add(did)
            calls[key] += 1
    for key, cs in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        print(f"{key[0]} [{key[1]} WGs] x{calls[key]}")
        wc = cs.get("SQ_WAVE_CYCLES", 0)
        for k, v in sorted(cs.items()):
            extra = f"   /WAVE_CYCLES = {v / wc:.3f}" if wc and k != "SQ_WAVE_CYCLES" else ""
            print(f"    {k:28s} {v:.4g}{extra}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
