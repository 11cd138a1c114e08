#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of tools/small_trace.py: per kernel the median duration and the median idle gap
before it (end of the previous kernel -> its start), over the last 100 steps."""
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-300:]
by = {}
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0][-60:]
    d = by.setdefault(n, ([], []))
    d[0].append((e - s) / 1e3)
    if prev_end is not None: d[1].append((s - prev_end) / 1e3)
    prev_end = e
tot = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3 / (len(rows) / len(by))
for n, (du, gp) in by.items():
    print("%-62s n=%3d  dur %.1f us  gap before %.1f us" % (n, len(du), statistics.median(du), statistics.median(gp) if gp else 0))
print("per step: %.1f us" % tot)
