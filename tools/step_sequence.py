"""One training step of a rocprofv3 --kernel-trace CSV as an ordered list, per hardware queue: every launch of the last
complete step (steps are delimited by adam_kernel on the training queue) with its start offset, duration and the idle gap
on its queue before it.  Shows where on the critical queue the time between the big kernels goes."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey],
              r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]) for r in rows))
adam = [e for e in ev if e[3].startswith("adam_kernel")]
# (two optimizer launches per step since round 5 - gated and ungated tensors: the LAST one of a cluster ends the step)
adam = [e for i, e in enumerate(adam) if i + 1 == len(adam) or adam[i + 1][0] - e[1] > 500_000]
t0, t1 = adam[-2][1], adam[-1][1]
main_q = adam[-1][2]
step = [e for e in ev if t0 <= e[0] < t1]
print(f"step window {(t1 - t0) / 1e6:.3f} ms, {len(step)} launches; training queue = {main_q}")
per_q = defaultdict(list)
for e in step:
    per_q[e[2]].append(e)
for q, es in sorted(per_q.items(), key=lambda kv: kv[0] != main_q):
    busy = sum(e[1] - e[0] for e in es) / 1e3
    print(f"\nqueue {q}: {len(es)} launches, {busy:.0f} us busy")
    prev_end = None
    gaps = defaultdict(lambda: [0, 0.0])
    for b, e, _, name in es:
        gap = (b - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"  {(b - t0) / 1e3:9.1f}  +{gap:7.1f}  {(e - b) / 1e3:7.1f} us  {name}")
        g = gaps["<2" if gap < 2 else "2-5" if gap < 5 else "5-20" if gap < 20 else ">=20"]
        g[0] += 1
        g[1] += max(gap, 0.0)
        prev_end = max(e, prev_end or e)
    print("  gaps before a launch on this queue: " + ", ".join(f"{k} us: {v[0]} ({v[1]:.0f} us)" for k, v in gaps.items()))
