"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per dispatch."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
words = sys.argv[2].split(",") if len(sys.argv) > 2 else []
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
with open(path) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
        if words and not any(w in name for w in words):
            continue
        key = (name, r.get("Grid_Size", ""))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[key].add(r["Dispatch_Id"])
for key, ctrs in sorted(agg.items()):
    n = len(calls[key])
    print(key[0], "grid", key[1], "dispatches", n)
    for c, v in sorted(ctrs.items()):
        print(f"    {c:32s} {v / n:16.1f}")
