#!/usr/bin/env python3
"""Kernel-sum against wall of the denoise STEP in its steady state, from a rocprofv3 --kernel-trace csv of bench.py: the launches
between two consecutive p_sample_update kernels are one step; per step: span (end of the previous update to end of this one),
busy (sum of kernel durations) and what is left.      python tools/step_gap.py <kernel_trace.csv> [steps=15]"""
import csv
import statistics as st
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
ends = [i for i, r in enumerate(rows) if "p_sample_update" in r["Kernel_Name"]][-(n + 1):]
spans, busys, counts = [], [], []
for a, b in zip(ends[:-1], ends[1:]):
    ks = rows[a + 1:b + 1]
    spans.append((int(ks[-1]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3)
    busys.append(sum(int(k["End_Timestamp"]) - int(k["Start_Timestamp"]) for k in ks) / 1e3)
    counts.append(len(ks))
sp, bu = st.median(spans), st.median(busys)
print(f"steps {len(spans)}  launches per step {counts[0]}  step span (median) {sp:.1f} us  kernel sum {bu:.1f} us  "
      f"idle between kernels {sp - bu:.1f} us = {100 * (1 - bu / sp):.2f} %  ({(sp - bu) / counts[0]:.3f} us per launch)")
per = {}
for a, b in zip(ends[:-1], ends[1:]):
    for k in rows[a + 1:b + 1]:
        nm = k["Kernel_Name"].replace("(anonymous namespace)::", "").replace("ditto::", "").replace("void ", "").split("(")[0][:56]
        e = per.setdefault(nm, [0, 0.0])
        e[0] += 1
        e[1] += (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e3
for nm, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {c / len(spans):6.1f} launches/step  {t / c:7.1f} us each  {t / len(spans):8.1f} us/step  {nm}")
