"""profiles/profile_summary.json from a rocprofv3 `--kernel-trace --stats` run of bench.py: per kernel family the time and
launches per step, tagged with the fingerprint of the kernel sources (bench.py reports `profile_frac` from it only when the
fingerprint matches the sources it runs).  usage: profile_summary.py kernel_stats.csv STEPS_TRACED [out.json]"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def category(n):
    if 'spconv_fwd' in n or 'spconv_tiles' in n or 'spconv_msplit' in n: return 'conv fwd/dgrad'
    if 'wgrad' in n: return 'wgrad (+reduce)'
    if 'bn_' in n: return 'batchnorm'
    return None


def main():
    from bench import kernel_source_fingerprint
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = float(sys.argv[2])
    out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "profile_summary.json")
    fam = defaultdict(lambda: [0.0, 0])
    total = [0.0, 0]
    for r in rows:
        t, c = float(r["TotalDurationNs"]), int(r["Calls"])
        total[0] += t
        total[1] += c
        k = category(r["Name"])
        if k:
            fam[k][0] += t
            fam[k][1] += c
    rec = {"kernel_source_fingerprint": kernel_source_fingerprint(), "steps_traced": steps,
           "command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-validation",
           "families": {k: {"ms_per_step": v[0] / 1e6 / steps, "launches_per_step": v[1] / steps} for k, v in fam.items()},
           "all_kernels": {"ms_per_step": total[0] / 1e6 / steps, "launches_per_step": total[1] / steps}}
    with open(out_path, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
