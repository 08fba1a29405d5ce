#!/bin/bash
# Round-4 measurement run: tools/final_measure.sh (GPU tests, default bench line, rocprofv3 stats + PMC passes behind
# profiles/profile_summary.json / traffic.json, per-kernel rooflines, host / critical-path tools, other BASELINE configs) plus
# what round 4 added: blocking vs device-counted proposal stage (interleaved A/B), the eval path, the config-4 profile.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
bash "$R/tools/final_measure.sh" > /dev/null 2>&1
O=$R/gpurun_out/final
cd "$R"
run() { env "$@" timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3))"; }
for r in 1 2 3 4; do run GPN_PROPOSALS_SYNC=1; run GPN_PROPOSALS_SYNC=0; done > "$O/sync_free_ab.txt" 2>&1
timeout 200 python tools/eval_bench.py 2>/dev/null | tail -1 > "$O/bench_eval_batch4.json"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p50
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p50 -o t -- python "$R/bench.py" --points 50000 --batch 4 --steps 16 --warmup 4 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
f=$(find /tmp/p50 -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "$O/points50000batch4_kernel_stats.csv"; python "$R/tools/gpu_categories.py" "$f" 21 > "$O/points50000batch4_gpu_time_by_category.txt" 2>&1 < /dev/null; fi
cd "$R"; tail -2 "$O/pytest_gpu.txt"; cut -c1-200 "$O/bench_default.json"; cat "$O/sync_free_ab.txt"; cat "$O/gpu_time_by_category.txt" | head -4
