#!/usr/bin/env python3
"""Per-kernel LDS utilisation from one rocprofv3 counter pass over bench.py
(`--pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE`, no trace flags).
LDS-active = SQ_LDS_IDX_ACTIVE (summed over the CUs) / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs), per-dispatch averages;
conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
    python tools/pmc_lds.py <pmc_csv> > profiles/rNN_lds.txt"""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ditto::", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
print(__doc__.split("\n")[1], __doc__.split("\n")[2])
rows = []
for k, c in acc.items():
    if "SQ_LDS_IDX_ACTIVE" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    n = len(c["GRBM_GUI_ACTIVE"])
    idx = sum(c["SQ_LDS_IDX_ACTIVE"]) / n
    bank = sum(c.get("SQ_LDS_BANK_CONFLICT", [0])) / n
    addr = sum(c.get("SQ_LDS_ADDR_CONFLICT", [0])) / n
    act = sum(c["GRBM_GUI_ACTIVE"]) / n / 8.0
    us = sum(dur[k]) / max(len(dur[k]), 1)
    rows.append((idx / (act * 256.0) if act else 0.0, bank / idx if idx else 0.0, addr / idx if idx else 0.0, n, us, k))
for frac, bank, addr, n, us, k in sorted(rows, reverse=True):
    if us >= 20.0:
        print(f"  {100 * frac:5.1f}% LDS-active  bank-conflict {100 * bank:4.1f}%  addr-conflict {100 * addr:4.1f}%  n={n:4d}  {us:8.1f} us  {k[:100]}")
