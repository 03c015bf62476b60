#!/usr/bin/env python3
"""Print a rocprofv3 --kernel-trace --stats kernel_stats.csv as: calls, average us, total ms, short kernel name.
    python tools/kstats.py <dir or csv> [name filter regex]"""
import csv, glob, os, re, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)[0]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for r in csv.DictReader(open(path)):
    n = r["Name"]
    if pat and not pat.search(n):
        continue
    n = re.sub(r"\(anonymous namespace\)::|ditto::|void ", "", n)
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f} us {float(r['TotalDurationNs']) / 1e6:9.2f} ms  {n[:100]}")
