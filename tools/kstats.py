#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: tools/kstats.py <dir-or-csv>."""
import csv
import glob
import sys

path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(path + "/**/*kernel_stats.csv", recursive=True)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(anonymous namespace)::")[-1].split("(")[0][:44]
        print(f"{name:46s} calls={r['Calls']:>5s} avg={float(r['AverageNs']) / 1e3:8.1f}us "
              f"min={float(r['MinNs']) / 1e3:8.1f} max={float(r['MaxNs']) / 1e3:8.1f}")
