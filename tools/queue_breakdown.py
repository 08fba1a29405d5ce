"""Per hardware queue of a rocprofv3 --kernel-trace CSV: kernel time per step and the top kernels - which stream is the long one."""
import csv
import sys
from collections import defaultdict

path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
per_q = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
span = {}
for r in rows:
    q = r[qkey]
    b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    per_q[q][name][0] += 1
    per_q[q][name][1] += (e - b) / 1e3
    lo, hi = span.get(q, (b, e))
    span[q] = (min(lo, b), max(hi, e))
for q, ks in sorted(per_q.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
    total = sum(v[1] for v in ks.values())
    n = sum(v[0] for v in ks.values())
    print(f"queue {q}: {total / steps / 1e3:.2f} ms of kernels per step, {n / steps:.0f} launches per step")
    for name, (c, t) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"    {t / steps:8.1f} us/step {c / steps:7.1f} x  {name}")
