#!/usr/bin/env python3
"""Summarise tools/pmc_attn.sh output: per counter, the average over the attention kernel's dispatches.
    python tools/pmc_attn_sum.py gpurun_out/pmc_attn_<flags>"""
import collections, csv, glob, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn64" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:32s} {sum(v)/len(v):16.0f}   (n={len(v)})")
