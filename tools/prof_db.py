#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 results .db (kernel name, launches, total and average duration)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from {kd} d "
                   f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total {tot / 1e3:.2f} ms over {sum(r[1] for r in rows)} launches")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{r[2] / tot * 100:5.1f}% {r[1]:6d} x {r[3]:9.1f} us  {r[0][:120]}")

# optional third argument: a kernel-name substring -> that kernel's dispatches grouped by grid size
if len(sys.argv) > 3:
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    gx = [c for c in cols if c.lower() in ("grid_size_x", "grid_x", "grid_size")][0]
    gy = [c for c in cols if c.lower() in ("grid_size_y", "grid_y")]
    sel = f"d.{gx}" + (f", d.{gy[0]}" if gy else "")
    print(f"-- '{sys.argv[3]}' by grid ({sel})")
    for r in cur.execute(f"select {sel}, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id "
                         f"where s.kernel_name like ? group by {sel} order by 1", (f"%{sys.argv[3]}%",)):
        print("   grid", r[:-2], f"{r[-2]:5d} x {r[-1]:9.1f} us")
