"""Forward vs backward launches of the conv / BatchNorm kernels on the training queue of a rocprofv3 --kernel-trace CSV: the
same kernel template computes a layer's forward and its dgrad, so the two averages differ only by what surrounds the launch
(the weight-gradient stream beside the backward pass, the BatchNorm-sum epilogues, accumulate mode).  Last complete step."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey],
              r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]) for r in rows))
adam = [e for e in ev if e[3].startswith("adam_kernel")]
# (two optimizer launches per step since round 5 - gated and ungated tensors: the LAST one of a cluster ends the step)
adam = [e for i, e in enumerate(adam) if i + 1 == len(adam) or adam[i + 1][0] - e[1] > 500_000]
t0, t1, main_q = adam[-2][1], adam[-1][1], adam[-1][2]
step = [e for e in ev if t0 <= e[0] < t1 and e[2] == main_q]
tb = min(e[0] for e in step if "bwd" in e[3])
f, b = defaultdict(list), defaultdict(list)
for s, e, _, n in step:
    if n.startswith("spconv") or n.startswith("bn_"):
        (f if s < tb else b)[n].append((e - s) / 1e3)
tf = tbw = 0.0
for n in sorted(set(f) | set(b)):
    fa, ba = f.get(n, []), b.get(n, [])
    tf += sum(fa)
    tbw += sum(ba)
    print(f"{n:46s} fwd {len(fa):3d} x {sum(fa) / max(len(fa), 1):6.1f} us   bwd {len(ba):3d} x {sum(ba) / max(len(ba), 1):6.1f} us")
conv_f = sum(sum(v) for k, v in f.items() if k.startswith("spconv"))
conv_b = sum(sum(v) for k, v in b.items() if k.startswith("spconv"))
print(f"conv launches: forward {conv_f:.0f} us, backward {conv_b:.0f} us; conv + BatchNorm: forward {tf:.0f} us, backward {tbw:.0f} us")
