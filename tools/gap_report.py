#!/usr/bin/env python3
"""Idle time on the GPU between kernels of a rocprofv3 --kernel-trace run: total, and the largest gaps grouped by the kernel that
follows them.      python tools/gap_report.py <dir or *_kernel_trace.csv> [skip_first_n_kernels]"""
import csv, glob, os, re, sys, collections
path = sys.argv[1]
if os.path.isdir(path):
    path = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))[skip:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = collections.defaultdict(lambda: [0, 0])
end = int(rows[0]["End_Timestamp"])
for r in rows[1:]:
    s = int(r["Start_Timestamp"])
    g = s - end
    if g > 0:
        n = re.sub(r"\(anonymous namespace\)::|ditto::|void ", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n)[:70]
        gaps[n][0] += g; gaps[n][1] += 1
    end = max(end, int(r["End_Timestamp"]))
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {(span - busy) / 1e6:.2f} ms")
for n, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"  {g / 1e3:9.1f} us idle in {c:5d} gaps (avg {g / c / 1e3:6.2f} us) before  {n}")
