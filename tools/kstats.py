#!/usr/bin/env python3
"""print the top rows of a rocprofv3 `--stats` kernel CSV: name, calls, total ms, average us, share"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    t = float(r["TotalDurationNs"])
    print("%-72s %7s %9.2f ms %8.1f us %6.2f %%" % (r["Name"][:72], r["Calls"], t / 1e6, float(r["AverageNs"]) / 1e3, 100 * t / tot))
print("total %.1f ms" % (tot / 1e6))
