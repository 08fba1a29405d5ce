#!/bin/bash
# the three profiler passes behind bench.py's profile_frac / traffic (profiles/profile_summary.json, profiles/traffic.json) + a
# smoke and executor check: what has to be re-run after ANY edit under gapartnet_amd/csrc (kernel-source fingerprint), when the
# full tools/final_measure.sh does not fit the GPU budget.  ~45 s on the box.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/mini
mkdir -p "$O"
export TMPDIR=/tmp
(cd "$R" && timeout 40 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.txt" 2>&1 < /dev/null); tail -1 "$O/smoke.txt"
cd /tmp
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o t -- python "$R/bench.py" --steps 16 --warmup 4 --no-cpu-baseline --no-validation > "$O/bench_under_rocprof.log" 2>&1 < /dev/null
f=$(find /tmp/p1 -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "$O/kernel_stats.csv"; python "$R/tools/gpu_categories.py" "$f" 21 > "$O/gpu_time_by_category.txt" 2>&1 < /dev/null; (cd "$R" && python tools/profile_summary.py "$f" 21 "$O/profile_summary.json" > /dev/null 2>&1 < /dev/null); fi
f=$(find /tmp/p1 -name "*kernel_trace.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then python "$R/tools/step_sequence.py" "$f" > "$O/step_sequence.txt" 2>&1 < /dev/null; python "$R/tools/queue_breakdown.py" "$f" 21 > "$O/queue_breakdown.txt" 2>&1 < /dev/null; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 60 rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -o t -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-validation > /dev/null 2>&1 < /dev/null
  f=$(find /tmp/p_$c -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python "$R/tools/pmc_summary.py" "$f" spconv > "$O/pmc_$c.txt" 2>&1 < /dev/null; fi
done
f1=$O/pmc_FETCH_SIZE.txt; f2=$O/pmc_WRITE_SIZE.txt
if [ -s "$f1" ] && [ -s "$f2" ]; then (cd "$R" && python tools/pmc_traffic.py "$f1" "$f2" "$O/traffic.json" spconv_fwd,spconv_tiles,spconv_msplit > /dev/null 2>&1); fi
cat "$O/gpu_time_by_category.txt" | tail -3; cat "$O/traffic.json" | grep fingerprint
timeout 40 python -m pytest "$R/tests" -m gpu -q -x -k "paired_passes or native_executor" 2>&1 < /dev/null | tail -1
