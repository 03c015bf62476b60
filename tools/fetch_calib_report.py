#!/usr/bin/env python3
"""Sum up tools/probe_fetch_calib.hip's two counter passes: counted bytes / moved bytes per access pattern.
    python tools/fetch_calib_report.py <fetch_counter_collection.csv> <write_counter_collection.csv>"""
import collections
import csv
import sys

GiB = float(1 << 30)
MOVED = {"read_x4": GiB, "read_x2": GiB, "read_rows_x2": (1 << 30) // 1536 // 128 * 128 * 1536, "read_rows_x4": (1 << 30) // 3072 // 128 * 128 * 3072,
         "dma_linear": GiB, "dma_rows": (1 << 30) // 1536 // 32 * 32 * 1536, "write_x4": GiB, "write_x2": GiB, "write_x4_nt": GiB}


def avg(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v[1:]) / max(len(v) - 1, 1) for k, v in acc.items()}      # first launch dropped (cold)


f, w = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
print("pattern          moved MB   FETCH_SIZE MB (x of moved)   WRITE_SIZE MB (x of moved)")
for k, moved in MOVED.items():
    fk, wk = f.get(k, 0.0) * 1024, w.get(k, 0.0) * 1024
    print(f"{k:14s} {moved / 1e6:9.1f}   {fk / 1e6:9.1f} ({fk / moved:5.3f})          {wk / 1e6:9.1f} ({wk / moved:5.3f})")
