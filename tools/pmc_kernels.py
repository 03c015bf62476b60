#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc pass (any counters): duration, each counter, and — when GRBM_GUI_ACTIVE is among
them — the effective clock GRBM_GUI_ACTIVE / 8 XCDs / duration.     python tools/pmc_kernels.py <counter_collection.csv> [name filter]"""
import collections
import csv
import sys

flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ditto::", "").replace("ditto::", "")
    if flt and flt not in k:
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
for k, c in sorted(acc.items()):
    us = sum(dur[k].values()) / len(dur[k])
    line = f"{k[:70]:70s} n={len(dur[k]):4d}  {us:8.1f} us"
    for name, vals in sorted(c.items()):
        line += f"  {name} {sum(vals) / len(vals):.4g}"
    if "GRBM_GUI_ACTIVE" in c:
        line += f"  clock ~{sum(c['GRBM_GUI_ACTIVE']) / len(c['GRBM_GUI_ACTIVE']) / 8 / us / 1e3:.2f} GHz"
    print(line)
