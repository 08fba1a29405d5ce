"""Summarise a rocprofv3 kernel_trace CSV: per (kernel, grid) launch count and mean duration."""
import csv
import sys
from collections import defaultdict

path, out = sys.argv[1], sys.argv[2]
words = sys.argv[3].split(",") if len(sys.argv) > 3 else ["spconv", "wgrad"]  # kernel-name filter
agg = defaultdict(lambda: [0, 0.0])
with open(path) as fh:
    rd = csv.DictReader(fh)
    for r in rd:
        name = r["Kernel_Name"]
        if not any(w in name for w in words):
            continue
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        key = (short, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""),
               r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[key][0] += 1
        agg[key][1] += d
with open(out, "w") as fh:
    fh.write("kernel,grid_x,grid_y,grid_z,lds,vgpr,calls,avg_us,total_us\n")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fh.write(",".join(map(str, k)) + f",{n},{t / n:.1f},{t:.0f}\n")
