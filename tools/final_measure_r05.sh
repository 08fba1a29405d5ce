#!/bin/bash
# Round-5 measurement run: tools/final_measure.sh (GPU tests, default bench line, rocprofv3 stats + PMC passes behind
# profiles/profile_summary.json / traffic.json, per-kernel rooflines, host / critical-path tools, other BASELINE configs) plus
# what round 5 added: the evaluation path, the disk-fed loop (packed cache / .pth workers / per-scene CPU preparation).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
bash "$R/tools/final_measure.sh" > /dev/null 2>&1
O=$R/gpurun_out/final
cd "$R"
for i in 1 2; do timeout 200 python tools/eval_bench.py 2>/dev/null | tail -1 > "$O/bench_eval_batch4_run$i.json"; done
timeout 600 python tools/pth_loader_bench.py 2> "$O/pth_loader.err" | tail -1 > "$O/pth_loader.json"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pe
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o t -- python "$R/tools/eval_bench.py" > /dev/null 2>&1 < /dev/null
f=$(find /tmp/pe -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "$O/eval_kernel_stats.csv"; python "$R/tools/gpu_categories.py" "$f" 1 > "$O/eval_gpu_time_by_category.txt" 2>&1 < /dev/null; fi
cd "$R"; tail -2 "$O/pytest_gpu.txt"; cut -c1-400 "$O/bench_default.json"; cat "$O/bench_eval_batch4_run1.json" | cut -c1-300; cut -c1-600 "$O/pth_loader.json"; head -4 "$O/gpu_time_by_category.txt"
