#!/usr/bin/env python3
"""Dispatches of a rocprofv3 --kernel-trace database grouped by (kernel, grid): calls, total and average duration.
    python tools/rocprof_by_grid.py <results.db> [top]"""
import sqlite3, sys
db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, grid_x/workgroup_x, grid_y/workgroup_y, workgroup_x, count(*), sum(end-start), avg(end-start) from kernels "
                   "group by name, grid_x, grid_y, workgroup_x order by 6 desc").fetchall()
tot = sum(r[5] for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(r[4] for r in rows)} dispatches")
for r in rows[:top]:
    print(f"{r[5] / 1e3:10.1f} us {100.0 * r[5] / tot:5.1f}%  calls {r[4]:5d}  avg {r[6] / 1e3:8.1f} us  wgs {r[1]:6d}x{r[2]:<3d} wg {r[3]:4d}  {r[0][:90]}")
