#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per (kernel name, grid size)."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60] + " grid=" + r.get("Grid_Size", "?") + " wg=" + r.get("Workgroup_Size", "?")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if k.startswith("void at::") or "elementwise" in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
