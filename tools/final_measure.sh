#!/bin/bash
# Round-end measurement on the GPU box: GPU tests, default bench line, per-kernel rooflines, rocprofv3 kernel stats of the
# bench (-> profiles/profile_summary.json for bench.py's profile_frac), conv kernel sweeps, and the PMC passes behind
# profiles/traffic.json.  Every step is bounded by `timeout`; nothing reads stdin.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 600 python -m pytest "$R/tests" -m gpu -q > "$O/pytest_gpu.txt" 2>&1 < /dev/null; tail -2 "$O/pytest_gpu.txt"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o t -- python "$R/bench.py" --steps 16 --warmup 4 --no-cpu-baseline --no-validation > "$O/bench_under_rocprof.log" 2>&1 < /dev/null
f=$(find /tmp/p1 -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "$O/kernel_stats.csv"; python "$R/tools/gpu_categories.py" "$f" 21 > "$O/gpu_time_by_category.txt" 2>&1 < /dev/null; (cd "$R" && python tools/profile_summary.py "$f" 21 "$O/profile_summary.json" > /dev/null 2>&1 < /dev/null); fi
f=$(find /tmp/p1 -name "*kernel_trace.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then python "$R/tools/step_sequence.py" "$f" > "$O/step_sequence.txt" 2>&1 < /dev/null; python "$R/tools/fwd_bwd_split.py" "$f" > "$O/fwd_bwd_split.txt" 2>&1 < /dev/null; python "$R/tools/summarize_trace.py" "$f" "$O/conv_kernels_by_grid.csv" "spconv,wgrad,bn_,ccl_,linear" > /dev/null 2>&1 < /dev/null; python "$R/tools/timeline.py" "$f" "$O/timeline.txt" 0.12 > /dev/null 2>&1 < /dev/null; python "$R/tools/queue_breakdown.py" "$f" 21 > "$O/queue_breakdown.txt" 2>&1 < /dev/null; python "$R/tools/backward_tail.py" "$f" > "$O/backward_tail.txt" 2>&1 < /dev/null; fi
cp "$O/profile_summary.json" "$R/profiles/profile_summary.json" 2>/dev/null
timeout 240 python "$R/bench.py" > "$O/bench_default.json" 2> "$O/bench_default.err" < /dev/null; cut -c1-260 "$O/bench_default.json"
for i in 1 2 3; do timeout 240 python "$R/bench.py" --no-cpu-baseline > "$O/bench_default_run$i.json" 2> /dev/null < /dev/null; done  # run-to-run spread on this box
timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/p_sq -o t -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-validation > /dev/null 2>&1 < /dev/null
f=$(find /tmp/p_sq -name "*counter_collection.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then python "$R/tools/pmc_summary.py" "$f" spconv > "$O/pmc_sq.txt" 2>&1 < /dev/null; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -o t -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-validation > /dev/null 2>&1 < /dev/null
  f=$(find /tmp/p_$c -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python "$R/tools/pmc_summary.py" "$f" spconv > "$O/pmc_$c.txt" 2>&1 < /dev/null; fi
done
f1=$O/pmc_FETCH_SIZE.txt; f2=$O/pmc_WRITE_SIZE.txt
if [ -s "$f1" ] && [ -s "$f2" ]; then (cd "$R" && python tools/pmc_traffic.py "$f1" "$f2" "$O/traffic.json" spconv_fwd,spconv_tiles,spconv_msplit > /dev/null 2>&1); fi
GPN_BENCH_FORCE_GRAD_SYNC=1 timeout 240 python "$R/bench.py" --no-cpu-baseline > "$O/bench_forced_grad_sync.json" 2> /dev/null < /dev/null
for cfg in "--schedule 5,10" "--points 50000 --batch 4" "--batch 32"; do
  tag=$(echo "$cfg" | tr -d ' -' | tr ',' '_')
  timeout 200 python "$R/bench.py" $cfg --no-cpu-baseline > "$O/bench_$tag.json" 2> /dev/null < /dev/null
done
for b in 4 16; do timeout 200 python "$R/bench.py" --batch $b --no-cpu-baseline > "$O/bench_batch$b.json" 2> /dev/null < /dev/null; done
timeout 200 python "$R/tools/kernel_rooflines.py" > "$O/kernel_rooflines.txt" 2>&1 < /dev/null; tail -3 "$O/kernel_rooflines.txt"
(cd "$R" && BENCH_LEVELS=5 timeout 200 python tools/conv_tiles_bench.py > "$O/conv_tiles_bench.txt" 2>&1 < /dev/null)
(cd "$R" && timeout 200 python tools/conv_split_sweep.py > "$O/conv_split.txt" 2>&1 < /dev/null)
(cd "$R" && timeout 200 python tools/sync_sites.py > "$O/sync_sites.txt" 2>&1 < /dev/null)
(cd "$R" && timeout 200 python tools/host_cprofile.py > "$O/host_cprofile.txt" 2>&1 < /dev/null)
(cd "$R" && timeout 200 python tools/critical_path.py > "$O/critical_path.txt" 2>&1 < /dev/null)
(cd "$R" && timeout 200 python tools/host_wait.py > "$O/host_wait.txt" 2>&1 < /dev/null)
(cd "$R" && timeout 200 python tools/backward_host.py > "$O/backward_host.txt" 2>&1 < /dev/null)
(cd "$R" && timeout 200 python tools/aten_sites.py > "$O/aten_sites.txt" 2>&1 < /dev/null)
(cd "$R" && timeout 200 python tools/step_times.py 300 > "$O/step_times.txt" 2>&1 < /dev/null)
for v in "" "--freeze-convs" "--no-bn-fuse"; do
  tag=$(echo "variant_default$v" | tr -d ' '); rm -rf /tmp/pv
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pv -o t -- python "$R/tools/step_loop.py" $v > /dev/null 2>&1 < /dev/null
  f=$(find /tmp/pv -name "*kernel_trace.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python "$R/tools/fwd_bwd_split.py" "$f" > "$O/$tag.txt" 2>&1 < /dev/null; fi
  (cd "$R" && timeout 200 python tools/step_loop.py $v 2>/dev/null | tail -1 >> "$O/$tag.txt")
done
ls -la "$O"
