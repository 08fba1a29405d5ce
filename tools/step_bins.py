"""Bin the last full training step of a rocprofv3 kernel trace (steps are delimited by the optimizer's
multi_tensor_apply kernels) into fixed time bins: busy fraction, launches and the dominant kernel of each bin."""
import csv
import sys
from collections import defaultdict

path, out = sys.argv[1], sys.argv[2]
bin_us = float(sys.argv[3]) if len(sys.argv) > 3 else 500.0
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
opt_idx = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r[2]]
# cluster optimizer launches (consecutive indices within 200 launches belong to one optimizer step)
clusters = []
for i in opt_idx:
    if clusters and i - clusters[-1][-1] < 200:
        clusters[-1].append(i)
    else:
        clusters.append([i])
a, b = clusters[-2][-1] + 1, clusters[-1][-1] + 1
step = rows[a:b]
t0 = step[0][0]
nb = int((step[-1][1] - t0) / 1e3 / bin_us) + 1
busy = [0.0] * nb
cnt = [0] * nb
names = [defaultdict(float) for _ in range(nb)]
for s, e, n in step:
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][:40]
    bi = int((s - t0) / 1e3 / bin_us)
    cnt[bi] += 1
    x = s
    while x < e:
        k = int((x - t0) / 1e3 / bin_us)
        edge = t0 + (k + 1) * bin_us * 1e3
        seg = min(e, edge) - x
        busy[k] += seg
        names[k][short] += seg
        x += seg
with open(out, "w") as fh:
    fh.write(f"step span {(step[-1][1] - t0) / 1e6:.2f} ms, {len(step)} launches, bins of {bin_us} us\n")
    for k in range(nb):
        top = sorted(names[k].items(), key=lambda kv: -kv[1])[:2]
        fh.write(f"{k * bin_us / 1e3:7.1f} ms  busy {busy[k] / (bin_us * 1e3):5.0%}  launches {cnt[k]:4d}  "
                 + ", ".join(f"{n} {t / 1e3:.0f}us" for n, t in top) + "\n")
