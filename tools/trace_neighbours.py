#!/usr/bin/env python3
"""Prints, for every launch of a kernel whose name contains PATTERN in the last part of a rocprofv3 kernel
trace, the launches just before and after it on the same queue (to see who issues it).

    python tools/trace_neighbours.py kernel_trace.csv PATTERN [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat, count = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 12
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]][-count:]
for i in idx:
    print("----")
    for j in range(max(0, i - 3), min(len(rows), i + 3)):
        r = rows[j]
        print("%s %-90s %8.1f us  grid %s  q%s" % ("=>" if j == i else "  ", r["Kernel_Name"][:90],
              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", "?"), r.get("Queue_Id", "?")))
