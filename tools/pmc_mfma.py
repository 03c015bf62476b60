#!/usr/bin/env python3
"""Per-kernel MFMA-busy fraction and effective clock from one rocprofv3 counter pass
(`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES`, no trace flags) over bench.py.
MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs * 1024), per-dispatch averages.
    python tools/pmc_mfma.py <pmc_csv> > profiles/rNN_mfma_busy.txt"""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ditto::", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
print("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -- python3 bench.py --steps 2 --warmup 1 "
      "--no-cpu-baseline --profile-steps 0 --loops 0 --no-sweep --no-c3")
print("MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs * 1024), per-dispatch "
      "averages; clock = GRBM_GUI_ACTIVE / 8 / dispatch duration (reads high on dispatches < 0.3 ms)")
rows = []
for k, c in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    n = len(c["GRBM_GUI_ACTIVE"])
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / n
    act = sum(c["GRBM_GUI_ACTIVE"]) / n / 8.0
    us = sum(dur[k]) / max(len(dur[k]), 1)
    rows.append((busy / (act * 1024.0) if act else 0.0, n, us, act / us / 1e3 if us else 0.0, k))
for frac, n, us, ghz, k in sorted(rows, reverse=True):
    if us >= 20.0:
        print(f"  {100 * frac:5.1f}% MFMA-busy  n={n:4d}  {us:8.1f} us  ~{ghz:4.2f} GHz  {k[:100]}")
