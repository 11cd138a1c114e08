#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc CSVs (one dir per pass) to mean counter values per kernel."""
import collections
import csv
import glob
import sys

out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "fa::" not in k:
            continue
        short = k.split("fa::")[1].split("<")[0]
        out[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(out):
    print(k)
    for c in sorted(out[k]):
        v = out[k][c]
        print("   %-32s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
