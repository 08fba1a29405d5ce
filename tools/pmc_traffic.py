"""HBM traffic per launch of the conv kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
runs of the same bench command, summarised by tools/pmc_summary.py) -> profiles/traffic.json, which bench.py reports
as roofline.traffic.  Units / corrections as MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950
FETCH_SIZE tallies 128-byte read requests at 64 bytes, so it is doubled; WRITE_SIZE is taken as is."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_fingerprint  # noqa: E402  (the sources the profiled .so was built from)


def aggregate(path, prefix):
    total, launches = 0.0, 0
    lines = open(path).read().splitlines()
    for i, line in enumerate(lines):
        if any(line.startswith(pre) for pre in prefix.split(",")):
            n = int(line.split("dispatches")[1])
            total += float(lines[i + 1].split()[1]) * n
            launches += n
    return total, launches


fetch_path, write_path, out = sys.argv[1], sys.argv[2], sys.argv[3]
prefix = sys.argv[4] if len(sys.argv) > 4 else "spconv_fwd,spconv_tiles,spconv_msplit"  # comma-separated kernel-name prefixes
f, n = aggregate(fetch_path, prefix)
w, n2 = aggregate(write_path, prefix)
res = {"kernel_prefix": prefix, "launches_profiled": n,
       "fetch_kib_per_launch_raw": f / n, "write_kib_per_launch": w / n2,
       "traffic_bytes_per_launch": (2.0 * f / n + w / n2) * 1024.0,
       "kernel_source_fingerprint": kernel_source_fingerprint(),
       "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B); separate --pmc passes of "
               "`python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-validation`"}
json.dump(res, open(out, "w"), indent=1)
print(res)
