"""Which stream ends a training step's backward pass?  From a rocprofv3 --kernel-trace CSV of bench.py: for every optimizer
launch (adam_kernel) the end of the last kernel of the training queue before it, the end of the last weight-gradient
kernel (other queue) before it, and the idle time of the training queue in between - i.e. how long the main chain waited
for the weight-gradient stream.  usage: backward_tail.py kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey], r["Kernel_Name"]) for r in rows))
adams = [e for e in ev if "adam_kernel" in e[3]]
# (two optimizer launches per step since round 5 - gated and ungated tensors: the FIRST one of a cluster follows the backward)
adams = [e for i, e in enumerate(adams) if i == 0 or e[0] - adams[i - 1][1] > 500_000]
main_q = adams[0][2]
out = []
for prev, a in zip(adams[1:], adams[2:]):
    before = [e for e in ev if e[1] <= a[0] and e[0] > prev[1]]  # (this step only: since the previous step's optimizer)
    main_prev = max((e for e in before if e[2] == main_q and "adam" not in e[3]), key=lambda e: e[1])
    wg = [e for e in before if "wgrad" in e[3] and e[2] != main_q]
    if not wg:
        continue
    wg_last = max(wg, key=lambda e: e[1])
    wg_first = min(wg, key=lambda e: e[0])
    busy = sum(e[1] - e[0] for e in wg)
    # the backward pass on the main queue: from the first wgrad launch to the optimizer
    main_busy = sum(e[1] - e[0] for e in before if e[2] == main_q and e[0] >= wg_first[0])
    out.append(((a[0] - main_prev[1]) / 1e3, (wg_last[1] - main_prev[1]) / 1e3, (a[0] - wg_first[0]) / 1e3, busy / 1e3, main_busy / 1e3,
                main_prev[3].split("(")[0][-40:]))
print("per step: main queue idle before the optimizer (us) | last wgrad end - last main kernel end (us) | backward window (us) |"
      " wgrad kernel time | main-queue kernel time inside the window | last main kernel")
for o in out:
    print(f"  {o[0]:8.1f} {o[1]:8.1f} {o[2]:9.1f} {o[3]:9.1f} {o[4]:9.1f}  {o[5]}")
n = len(out)
if n:
    print(f"mean: idle before optimizer {sum(o[0] for o in out) / n:.1f} us, wgrad tail past the main chain {sum(o[1] for o in out) / n:.1f} us, "
          f"backward window {sum(o[2] for o in out) / n:.1f} us, wgrad kernels {sum(o[3] for o in out) / n:.1f} us, main kernels {sum(o[4] for o in out) / n:.1f} us")
