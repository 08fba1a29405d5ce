#!/usr/bin/env python
"""per-kernel calls / average / time per step out of a rocprofv3 kernel_stats.csv, filtered by name substrings:
    python tools/kernel_table.py <kernel_stats.csv> <profiled steps> [substring ...]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
keys = sys.argv[3:]
total = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"]
    if keys and not any(k in n for k in keys):
        continue
    nm = re.sub(r"\(anonymous namespace\)::", "", n)
    nm = re.sub(r"^void ", "", nm)
    nm = re.split(r"\(", nm)[0][:60]
    per_step = float(r["TotalDurationNs"]) / 1e3 / steps
    total += per_step
    print(f"{nm:60s} {int(r['Calls']) / steps:6.1f} calls/step  avg {float(r['AverageNs']) / 1e3:7.1f} us  {per_step:8.1f} us/step")
print(f"{'total':60s} {total:42.1f} us/step")
