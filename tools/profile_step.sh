#!/bin/bash
# rocprofv3 kernel stats of the bench + per-category GPU time + host op counts; results under gpurun_out/$1
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o t -- python "$R/bench.py" --steps 16 --warmup 4 --no-cpu-baseline --no-validation > "$O/bench_under_rocprof.log" 2>&1 < /dev/null
f=$(find /tmp/p1 -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "$O/kernel_stats.csv"; python "$R/tools/gpu_categories.py" "$f" 21 > "$O/gpu_time_by_category.txt" 2>&1; fi
f=$(find /tmp/p1 -name "*kernel_trace.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then python "$R/tools/timeline.py" "$f" "$O/timeline.txt" 0.4 > /dev/null 2>&1; fi
cd "$R"
timeout 200 python tools/phase_profile.py > "$O/phase.txt" 2>&1
timeout 200 python tools/op_count.py > "$O/opcount.txt" 2>&1
timeout 200 python tools/sync_sites.py > "$O/sync.txt" 2>&1
cat "$O/gpu_time_by_category.txt"; head -12 "$O/timeline.txt"; cat "$O/phase.txt" | tail -24; tail -14 "$O/opcount.txt"; tail -14 "$O/sync.txt"
