#!/bin/bash
# quick A/B on the GPU box: [tests matching $1], two default bench lines, and the rocprofv3 per-kernel averages of one bench run
# usage: tools/quick_ab.sh "<pytest -k expression or empty>" <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_${2:-x}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
if [ -n "$1" ]; then timeout 900 python -m pytest "$R/tests" -m gpu -q -x -k "$1" > "$O/pytest.txt" 2>&1 < /dev/null; tail -3 "$O/pytest.txt"; fi
for i in 1 2; do timeout 200 python "$R/bench.py" --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value'],1), 'pc/s', round(d['ms_per_step'],3), 'ms')"; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o t -- python "$R/bench.py" --steps 16 --warmup 4 --no-cpu-baseline > "$O/rp.log" 2>&1 < /dev/null
f=$(find /tmp/pq -name "*kernel_stats.csv" | head -1)
cp "$f" "$O/kernel_stats.csv"; python "$R/tools/gpu_categories.py" "$f" 21 > "$O/gpu_time_by_category.txt" 2>&1; cat "$O/gpu_time_by_category.txt"
f=$(find /tmp/pq -name "*kernel_trace.csv" | head -1)
python "$R/tools/queue_breakdown.py" "$f" 21 > "$O/queue_breakdown.txt" 2>&1
python "$R/tools/summarize_trace.py" "$f" "$O/by_grid.csv" "spconv,wgrad,bn_,ccl_" > /dev/null 2>&1
