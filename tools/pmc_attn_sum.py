#!/usr/bin/env python3
"""Summarise tools/pmc_attn.sh output: per attention kernel and counter, the average over the kernel's dispatches.
    python tools/pmc_attn_sum.py gpurun_out/pmc_attn_<flags | bwd>"""
import collections, csv, glob, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(attn\w*kernel(?:<[^>]*>)?)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kern in sorted(acc):
    print(kern)
    for k in sorted(acc[kern]):
        v = acc[kern][k]
        print(f"  {k:32s} {sum(v)/len(v):16.0f}   (n={len(v)})")
