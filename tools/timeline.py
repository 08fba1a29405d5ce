"""GPU timeline summary from a rocprofv3 kernel_trace CSV: busy vs idle time inside a window, top kernels, gap sizes.

usage: timeline.py <kernel_trace.csv> <out.txt> [window_fraction=0.5]   (the window is the LAST fraction of the trace,
i.e. the steady-state steps of a bench run)"""
import csv
import sys
from collections import defaultdict

path, out = sys.argv[1], sys.argv[2]
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
w0 = t1 - (t1 - t0) * frac
win = [r for r in rows if r[0] >= w0]
busy, cur_s, cur_e = 0, win[0][0], win[0][1]
gaps = []
for s, e, _ in win[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = cur_e - win[0][0]
def shorten(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:90]


agg = defaultdict(lambda: [0, 0])
after = defaultdict(list)  # idle time between the end of a kernel and the start of the next launch
for i, (s, e, n) in enumerate(win):
    short = shorten(n)
    agg[short][0] += 1
    agg[short][1] += e - s
    if i + 1 < len(win):
        after[short].append(max(0, win[i + 1][0] - e))
with open(out, "w") as fh:
    fh.write(f"window {span / 1e6:.2f} ms, {len(win)} launches, busy {busy / 1e6:.2f} ms ({busy / span:.1%}), idle {(span - busy) / 1e6:.2f} ms\n")
    for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1e9)):
        sel = [g for g in gaps if lo * 1e3 <= g < hi * 1e3]
        fh.write(f"gaps {lo}-{hi} us: {len(sel)} totalling {sum(sel) / 1e6:.2f} ms\n")
    fh.write("kernel,calls,total_ms,avg_us,share_of_busy,median_gap_after_us,total_gap_after_ms\n")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        g = sorted(after[k]) or [0]
        small = [x for x in g if x < 200e3]
        fh.write(f"{k},{n},{t / 1e6:.3f},{t / n / 1e3:.1f},{t / busy:.1%},{g[len(g) // 2] / 1e3:.1f},{sum(small) / 1e6:.2f}\n")
