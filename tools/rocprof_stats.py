#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 --kernel-trace results database (rocpd SQLite) as CSV:
    python tools/rocprof_stats.py <results.db> <out.csv>"""
import csv, sqlite3, sys
db, out = sys.argv[1:3]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 3), round(100.0 * r[2] / tot, 2), r[4], r[5]])
print(f"{len(rows)} kernels -> {out}")
